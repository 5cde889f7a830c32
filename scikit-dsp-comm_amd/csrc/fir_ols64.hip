// fir_ols64.hip -- FFT overlap-save FIR in FLOAT64 for gfx950 (MI355X): complex128 signals (any taps) and float64 signals
// with real taps (two real tiles ride in one complex tile).
//
// Serves multirate_FIR.filter / .dn (multirate_helper.py:104-109, 121-127) for the callers whose arrays are float64 --
// NumPy's default dtype, and the arithmetic the reference itself computes in (lfilter promotes everything to float64).
// Without it a 1024-tap filter on complex128 data is 4096 FP64 flop per sample of direct form (6.6 ms per 2^26 samples);
// in the frequency domain it is ~90.
//
// Tile: N = 4096 complex128 points, 256 threads x 16 points in registers, N = 16 x 16 x 16:
//   n = 256 a + 16 b + c          k = k1 + 16 k2 + 256 k3            (all digits in [0, 16))
//   pass 1  thread t = (b, c):   DFT16 over a  -> k1, times W_4096^(t k1)      (powers of W_4096^t, built on the fly)
//   xchg 1  through LDS [k1][b][c] (one workgroup barrier)
//   pass 2  thread (k1, c):      DFT16 over b  -> k2, times W_256^(c k2)
//   xchg 2  through LDS [k1][k2][c] (inside one 16-lane group: wave-local, no barrier)
//   pass 3  thread (k1, k2):     DFT16 over c  -> k3;   multiply by H[k] / N (64 KiB table, L2-resident)
// and the mirror image back (decimation in time), so no bit reversal exists.  LDS image: 16 x (16 x 17) complex128 =
// 68 KiB (row pitch 17: every 16-lane access pattern used here lands on 16 distinct 16-byte bank groups) -> two
// persistent workgroups per CU.  V = 4096 - OV valid outputs per tile, OV = Ntaps-1 rounded up to 256 (<= 2048: longer
// filters are partitioned by capi.hip).  Algorithmic bytes: 32 B per complex128 sample, 16 B per float64 sample.
// Precision: float64 butterflies, twiddle powers by repeated multiplication (<= 15 products): 1e-14 of the peak.
#include "skdsp_internal.hpp"
#include <complex>
#include <vector>
#include <cmath>


namespace skdsp {

namespace {

typedef double2 cdd;

constexpr int kN64 = 4096;
constexpr int kPitch64 = 16 * 17;  // complex elements per k1 row of the LDS image

__device__ __forceinline__ cdd cadd(cdd a, cdd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cdd csub(cdd a, cdd b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cdd cmul(cdd a, cdd w) { return make_double2(fma(a.x, w.x, -a.y * w.y), fma(a.x, w.y, a.y * w.x)); }
__device__ __forceinline__ cdd cmulc(cdd a, cdd w) { return make_double2(fma(a.x, w.x, a.y * w.y), fma(a.y, w.x, -a.x * w.y)); }
// a * (-i) (forward) / a * (+i) (inverse)
template <bool INV> __device__ __forceinline__ cdd mul_mi(cdd a) { return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x); }

// a * W_16^M forward (W = exp(-2 pi i / 16)) or its conjugate (INV); M compile-time
template <int M, bool INV> __device__ __forceinline__ cdd tw16(cdd a)
{
    constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, R2 = 0.70710678118654752440;
    constexpr int m = M & 15;
    if constexpr (m == 0) return a;
    else if constexpr (m == 4) return mul_mi<INV>(a);
    else if constexpr (m == 8) return make_double2(-a.x, -a.y);
    else if constexpr (m == 12) return mul_mi<!INV>(a);
    else {
        constexpr double c = (m == 1 || m == 15) ? C1 : (m == 2 || m == 14) ? R2 : (m == 3 || m == 13) ? S1 : (m == 5 || m == 11) ? -S1
                             : (m == 6 || m == 10) ? -R2 : -C1;                                   // cos(2 pi m / 16)
        constexpr double s = (m == 1 || m == 7) ? S1 : (m == 2 || m == 6) ? R2 : (m == 3 || m == 5) ? C1 : (m == 9 || m == 15) ? -S1
                             : (m == 10 || m == 14) ? -R2 : -C1;                                  // sin(2 pi m / 16)
        const cdd w = make_double2(c, INV ? s : -s);
        return cmul(a, w);
    }
}

template <bool INV> __device__ __forceinline__ void dft4(cdd x0, cdd x1, cdd x2, cdd x3, cdd &X0, cdd &X1, cdd &X2, cdd &X3)
{
    const cdd s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
    X0 = cadd(s02, s13);
    X2 = csub(s02, s13);
    const cdd r = mul_mi<INV>(d13);
    X1 = cadd(d02, r);
    X3 = csub(d02, r);
}

template <int I, int E, class F> __device__ __forceinline__ void static_for64(F &&f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for64<I + 1, E>(static_cast<F &&>(f));
    }
}

// 16-point DFTs IN PLACE (radix 4 x 4), so that a pass needs one 16-element register array instead of three:
//   dft16_f  natural order in, output X[k] at slot P16(k) = 4 (k & 3) + (k >> 2)   (decimation in frequency)
//   dft16_g  input Z[m] at slot P16(m), natural order out; unnormalised inverse      (decimation in time)
// P16 is its own inverse (the transpose of the 4 x 4 index grid) and every index below is a compile-time constant, so
// the permutation costs nothing: the forward passes hand their output to LDS / H through P16, the inverse pass 3 takes
// the spectrum exactly where the forward pass 3 left it.
__device__ __host__ constexpr int P16(int k) { return ((k & 3) << 2) | (k >> 2); }

template <bool INV> __device__ __forceinline__ void dft4_ip(cdd &x0, cdd &x1, cdd &x2, cdd &x3)
{
    const cdd s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
    const cdd r = mul_mi<INV>(d13);
    x0 = cadd(s02, s13);
    x2 = csub(s02, s13);
    x1 = cadd(d02, r);
    x3 = csub(d02, r);
}

__device__ __forceinline__ void dft16_f(cdd *v)
{
    // stage 1: DFT4 over n2 for each n1 (slots n1, n1 + 4, n1 + 8, n1 + 12): slot n1 + 4 k2 = a[n1][k2]
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) dft4_ip<false>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
    // twiddle W_16^(n1 k2), stage 2: DFT4 over n1 for each k2 (slots 4 k2 .. 4 k2 + 3): slot 4 k2 + k1 = X[k2 + 4 k1]
    static_for64<0, 4>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value;
        v[4 * k2 + 1] = tw16<k2, false>(v[4 * k2 + 1]);
        v[4 * k2 + 2] = tw16<2 * k2, false>(v[4 * k2 + 2]);
        v[4 * k2 + 3] = tw16<3 * k2, false>(v[4 * k2 + 3]);
    });
    #pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) dft4_ip<false>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
}

__device__ __forceinline__ void dft16_g(cdd *v)
{
    // input Z[m1 + 4 m2] at slot 4 m1 + m2.  stage A: inverse DFT4 over m2 for each m1 (slots 4 m1 .. 4 m1 + 3): slot 4 m1 + r
#pragma unroll
    for (int m1 = 0; m1 < 4; ++m1) dft4_ip<true>(v[4 * m1], v[4 * m1 + 1], v[4 * m1 + 2], v[4 * m1 + 3]);
    // conj twiddle W_16^(m1 r) on slot 4 m1 + r; stage B: inverse DFT4 over m1 for each r (slots r, r + 4, r + 8, r + 12): x[r + 4 q] at slot r + 4 q
    static_for64<1, 4>([&](auto mc) {
        constexpr int m1 = decltype(mc)::value;
        v[4 * m1 + 1] = tw16<m1, true>(v[4 * m1 + 1]);
        v[4 * m1 + 2] = tw16<2 * m1, true>(v[4 * m1 + 2]);
        v[4 * m1 + 3] = tw16<3 * m1, true>(v[4 * m1 + 3]);
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) dft4_ip<true>(v[r], v[r + 4], v[r + 8], v[r + 12]);
}

struct Ols64Args {
    const double *x;
    double *y;
    int64_t n, n_hist;
    const cdd *Hp;    // [16][256]: Hp[k3 * 256 + 16 k1 + k2] = H[k1 + 16 k2 + 256 k3] / N
    const cdd *W1;    // [256]: W_4096^t
    const cdd *W2;    // [16]:  W_256^c
    int ov, V, a0;    // a0 = ov / 256: first stored 256-block
    int64_t ntiles;
    int dec;
    unsigned dec_magic;   // ceil(2^32 / dec): (g * dec_magic) >> 32 = g / dec for the tile-local g < 2^14 of the decimating store
    int64_t n_keep;
    // up > 1 (multirate_FIR.up with long phases, as in fir_ols.hip): the walk runs over (tile, phase) pairs -- index w is input tile
    // w / up filtered with phase w % up (Hp: up tables of 4096 bins), output i of the pair lands at y[i * up + phase]
    int up;
    int64_t up_pitch;   // > 0: the phases as rows, y[phase * up_pitch + i] (scratch; interleave_launch weaves them)
    CarefulFir cf;      // the filter as the exact path of a poisoned tile reads it (careful.hpp)
};

// DEC: the decimating store (multirate_FIR.dn) is its own instantiation: the plain filter carries none of its code
// UP: multirate_FIR.up over (tile, phase) pairs; H of the pair's phase streamed from L2, stores with stride up (a thread's outputs are
//     lane-consecutive here, so every store instruction writes 64 consecutive outputs of the phase as it is)
// XR: a REAL signal into the complex tile (imaginary part zero) -- multirate_FIR.up of float64 signals with an even L runs its phases in pairs,
//     x * (h_2k + i h_2k+1) = y_2k + i y_2k+1: one complex pass whose output IS the interleaved pair as one 16-byte element (see fir_ols.hip)
template <bool REAL, bool DEC, bool UP = false, bool XR = false>
__global__ __launch_bounds__(256, 2) void ols64_tile_kernel(Ols64Args A)
{
    __shared__ cdd img[16 * kPitch64];
    __shared__ unsigned long long ols_noted;  // poisoned tiles, by walk step (careful.hpp)
    const int t = threadIdx.x;
    if (t == 0) ols_noted = 0;
    const int hi4 = t >> 4, lo4 = t & 15;
    const cdd w1 = A.W1[t];        // W_4096^t            (pass 1: t = 16 b + c)
    const cdd w2 = A.W2[lo4];      // W_256^c             (pass 2: thread (k1, c))
    // This thread's 16 bins of H: in registers for the whole launch in the complex kernel (0.691 vs 0.738 ms at 2^26), streamed
    // from L2 per tile in the two-real-tiles kernel (registers there: 0.487 vs 0.401 ms -- its loads and stores need more of them)
    constexpr bool HREG = !UP && !REAL;
    cdd hh[HREG ? 16 : 1];
    if (HREG) {
#pragma unroll
        for (int k3 = 0; k3 < (HREG ? 16 : 1); ++k3) hh[k3] = A.Hp[k3 * 256 + t];
    }

    // x[in0 + 256 a + t] -> dst[a] (complex), or (xA, xB) of two real tiles; zero outside [-n_hist, n)
    auto load_tile = [&](int64_t tile, int tl, cdd *dst) __attribute__((always_inline)) {
        if (REAL) {
            const int64_t inA = (2 * tile) * A.V - A.ov, inB = inA + A.V;
            const bool interior = inA >= -A.n_hist && inB + kN64 <= A.n;
            if (interior) {   // uniform base + 32-bit lane offset: SGPR-base addressing, no 64-bit address pair per access
                const double *pa = A.x + inA, *pb = A.x + inB;
#pragma unroll
                for (int a = 0; a < 16; ++a)
                    dst[a] = make_double2(__builtin_nontemporal_load(pa + (unsigned)(256 * a + tl)), __builtin_nontemporal_load(pb + (unsigned)(256 * a + tl)));
            } else {
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    const int64_t ga = inA + 256 * a + tl, gb = ga + A.V;
                    double re = 0.0, im = 0.0;
                    if (ga >= -A.n_hist && ga < A.n) re = A.x[ga];
                    if (gb >= -A.n_hist && gb < A.n) im = A.x[gb];
                    dst[a] = make_double2(re, im);
                }
            }
        } else if (XR) {
            const int64_t in0 = tile * A.V - A.ov;
            const bool interior = in0 >= -A.n_hist && in0 + kN64 <= A.n;
            if (interior) {
                const double *px = A.x + in0;
#pragma unroll
                for (int a = 0; a < 16; ++a) dst[a] = make_double2(__builtin_nontemporal_load(px + (unsigned)(256 * a + tl)), 0.0);
            } else {
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    const int64_t g = in0 + 256 * a + tl;
                    dst[a] = make_double2((g >= -A.n_hist && g < A.n) ? A.x[g] : 0.0, 0.0);
                }
            }
        } else {
            const int64_t in0 = tile * A.V - A.ov;
            const bool interior = in0 >= -A.n_hist && in0 + kN64 <= A.n;
            typedef double v2d_t __attribute__((ext_vector_type(2)));
            if (interior) {
                // PLAIN loads here (the two-real-tiles kernel keeps nontemporal ones; A/B on the same box, alternating): complex128
                // 0.587 -> 0.549 ms, float64 0.320 -> 0.325 ms.  (Neighbouring tiles share OV of their 4096 points -- a quarter at
                // 1024 taps -- and run on the same XCD at the same time; the FETCH_SIZE counter did NOT drop with the change, so
                // whatever helps is not fewer fabric reads.)
                const v2d_t *px = reinterpret_cast<const v2d_t *>(A.x) + in0;
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    const v2d_t q = px[(unsigned)(256 * a + tl)];
                    dst[a] = make_double2(q.x, q.y);
                }
            } else {
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    const int64_t g = in0 + 256 * a + tl;
                    dst[a] = (g >= -A.n_hist && g < A.n) ? make_double2(A.x[2 * g], A.x[2 * g + 1]) : make_double2(0.0, 0.0);
                }
            }
        }
    };
    int64_t tile = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
    auto tile_first = [&]() -> int64_t { return (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x; };
    // The next tile's samples are requested into registers behind the H product (with the running twiddle powers no longer hoisted the
    // kernels use 154 - 174 of their 256 registers: room for the 64 of a tile) and have the whole inverse transform to arrive; interior tiles
    // only.  Measured (2^26, 1024 taps, same box, alternating): float64 0.327 -> 0.308 ms (the round-4 build with its cache-line touches
    // through global_load_lds: 0.315), complex128 0.512 -> 0.507 (round 4, with 18 spilled registers: 0.542).
    constexpr bool PREF = true;
    auto interior_of = [&](int64_t tn) -> bool {
        const int64_t i0 = (REAL ? 2 * tn : tn) * A.V - A.ov;
        return i0 >= -A.n_hist && i0 + (REAL ? A.V : 0) + kN64 <= A.n;
    };
    cdd v[16];
    bool have = false;   // v holds the samples of `tile` (requested a tile ahead)
    for (; tile < A.ntiles; tile += gridDim.x) {
        // .up: pair index -> (input tile, phase); < 2^31 pairs (checked at launch)
        const int64_t tin = UP ? (int64_t)((unsigned)tile / (unsigned)A.up) : tile;
        const int ph = UP ? (int)((unsigned)tile % (unsigned)A.up) : 0;
        const cdd *Hq = UP ? A.Hp + (size_t)ph * kN64 : A.Hp;
        // opaque copies of the thread index: stop LICM from hoisting the 16 + 16 + 16 loop-invariant 64-bit addresses of the
        // loads, the H bins and the stores out of the tile loop (they were spilled and reloaded in front of every access)
        int tl = t, th = t, ts = t;
        asm volatile("" : "+v"(tl));
        if (!PREF || !have) load_tile(tin, tl, v);
        // ---- pass 1: DFT16 over a, twiddle W_4096^(t k1) (running power), write [k1][b][c] ----
        dft16_f(v);   // X[k1] at v[P16(k1)]
        {
            cdd w = w1;
            asm volatile("" : "+v"(w.x), "+v"(w.y));   // (the running powers are recomputed per tile: hoisted out of the tile loop they are 60 spilled registers)
            img[0 * kPitch64 + hi4 * 17 + lo4] = v[0];
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) {
                img[k1 * kPitch64 + hi4 * 17 + lo4] = cmul(v[P16(k1)], w);
                w = cmul(w, w1);
            }
        }
        __syncthreads();
        // ---- pass 2: thread (k1 = hi4, c = lo4): DFT16 over b, twiddle W_256^(c k2), write [k1][k2][c] (same 16-lane group) ----
#pragma unroll
        for (int b = 0; b < 16; ++b) v[b] = img[hi4 * kPitch64 + b * 17 + lo4];
        dft16_f(v);
        {
            cdd w = w2;
            asm volatile("" : "+v"(w.x), "+v"(w.y));
            img[hi4 * kPitch64 + 0 * 17 + lo4] = v[0];
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) {
                img[hi4 * kPitch64 + k2 * 17 + lo4] = cmul(v[P16(k2)], w);
                w = cmul(w, w2);
            }
        }
        // ---- pass 3: thread (k1 = hi4, k2 = lo4): DFT16 over c; multiply by H ----
        // (H streamed from L2 -- the two-real-tiles kernel -- is requested HERE, a whole DFT16 ahead of its use: requested at
        // the multiply, behind the opaque copy of the thread index, every tile waited out an L2 round trip)
        constexpr bool EARLY = !HREG && !DEC && !(UP && !REAL);   // (the decimating-store and the complex .up instantiations have no registers to spare for it)
        if (EARLY) asm volatile("" : "+v"(th));
        cdd hs[HREG ? 1 : 16];
        if (EARLY) {
#pragma unroll
            for (int k3 = 0; k3 < 16; ++k3) hs[HREG ? 0 : k3] = Hq[k3 * 256 + th];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = img[hi4 * kPitch64 + lo4 * 17 + c];
        dft16_f(v);   // Z[k3] at v[P16(k3)]
        if (!EARLY) {
            asm volatile("" : "+v"(th));
            if (!HREG) {
#pragma unroll
                for (int k3 = 0; k3 < 16; ++k3) hs[HREG ? 0 : k3] = Hq[k3 * 256 + th];
            }
        }
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) v[P16(k3)] = cmul(v[P16(k3)], HREG ? hh[HREG ? k3 : 0] : hs[HREG ? 0 : k3]);
        cdd nx[PREF ? 16 : 1];
        bool pre = false;
        if constexpr (PREF) {
            const int64_t nw = tile + gridDim.x;
            const int64_t nt = UP ? (int64_t)((unsigned)nw / (unsigned)A.up) : nw;
            pre = nw < A.ntiles && interior_of(nt);
            if (pre) {
                int tp = t;
                asm volatile("" : "+v"(tp));
                load_tile(nt, tp, nx);
            }
        }
        // ---- inverse pass 3: over k3 -> c (takes the spectrum where it lies); the conj twiddle W_256^(c k2) is applied by the reader ----
        dft16_g(v);   // v[c] for thread (k1, k2)
#pragma unroll
        for (int c = 0; c < 16; ++c) img[hi4 * kPitch64 + lo4 * 17 + c] = v[c];
        // thread (k1, c = lo4) reads over k2, applies conj W_256^(c k2), inverse DFT16 over k2 -> b
        {
            cdd w = w2;
            asm volatile("" : "+v"(w.x), "+v"(w.y));
            v[0] = img[hi4 * kPitch64 + 0 * 17 + lo4];
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) {
                v[P16(k2)] = cmulc(img[hi4 * kPitch64 + k2 * 17 + lo4], w);
                w = cmul(w, w2);
            }
        }
        dft16_g(v);   // v[b] for thread (k1, c)   (rows 4 wave .. 4 wave + 3 belong to this wave alone until here)
#pragma unroll
        for (int b = 0; b < 16; ++b) img[hi4 * kPitch64 + b * 17 + lo4] = v[b];
        __syncthreads();
        // thread t = (b = hi4, c = lo4) reads over k1, conj W_4096^(t k1), inverse DFT16 over k1 -> a
        {
            cdd w = w1;
            asm volatile("" : "+v"(w.x), "+v"(w.y));   // (the running powers are recomputed per tile: hoisted out of the tile loop they are 60 spilled registers)
            v[0] = img[0 * kPitch64 + hi4 * 17 + lo4];
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) {
                v[P16(k1)] = cmulc(img[k1 * kPitch64 + hi4 * 17 + lo4], w);
                w = cmul(w, w1);
            }
        }
        dft16_g(v);   // v[a] = y[256 a + t]
        // "these are the results, in these registers, now" (LABNOTES R4.3): without it hipcc carries the finished transform in a form of its own
        // into the store section -- spilled VGPRs 66 -> 30 (complex, decimating), 22 -> 18 (complex), 4 -> 0 (real, decimating)
#pragma unroll
        for (int b = 0; b < 16; b += 4)
            asm volatile("" : "+v"(v[b].x), "+v"(v[b].y), "+v"(v[b + 1].x), "+v"(v[b + 1].y), "+v"(v[b + 2].x), "+v"(v[b + 2].y), "+v"(v[b + 3].x), "+v"(v[b + 3].y));
        asm volatile("" : "+v"(ts));
        if constexpr (PREF) {   // (the wait for the prefetch HERE, in front of the stores: vmcnt retires in order)
            if (pre) {
#pragma unroll
                for (int b = 0; b < 16; b += 4)
                    asm volatile("" ::"v"(nx[b].x), "v"(nx[b].y), "v"(nx[b + 1].x), "v"(nx[b + 1].y), "v"(nx[b + 2].x), "v"(nx[b + 2].y), "v"(nx[b + 3].x), "v"(nx[b + 3].y) : "memory");
            }
        }
        // ---- store the last V points ----
        // whole tile(s) inside the signal, no decimation: one copy of the 16 - a0 unguarded stores per possible a0
        // (compile-time offsets, no predicates -- as in fir_ols.hip: a run-time a0 made hipcc keep sixteen (exec mask, 64-bit
        // offset) pairs alive across the tile loop, and with them it spilled 72 VGPRs in the complex kernel: 0.691 -> 0.578 ms)
        const int64_t out0 = (REAL ? 2 * tin : tin) * A.V;
        const bool rows = UP && !DEC && A.up_pitch != 0;
        double *const ybase = rows ? A.y + (int64_t)ph * A.up_pitch * (REAL ? 1 : 2) : A.y;
        const bool full = !DEC && (!UP || rows) && out0 + (REAL ? 2 : 1) * (int64_t)A.V <= A.n;
        typedef double v2d_t __attribute__((ext_vector_type(2)));
        if (UP && !rows) {
            // y[(out0 + i) up + ph]: a uniform 64-bit base per 256-block plus a 32-bit per-lane byte offset (scalar-base stores)
            int a0 = A.a0;
            asm volatile("" : "+s"(a0));
            constexpr int ESZ = REAL ? 8 : 16;
            const int64_t left = A.n - out0;
            const bool whole = left >= (REAL ? 2 : 1) * (int64_t)A.V;   // (uniform: every tile but the last)
            const int lim = (left > (1 << 20) ? (1 << 20) : (int)left) - ts;   // outputs i < lim (relative to this lane's first) exist
            char *ua0 = reinterpret_cast<char *>(A.y) + (size_t)(out0 * A.up + ph) * ESZ;
            char *ub0 = ua0 + (size_t)A.V * A.up * ESZ;   // (the pair's second tile, two-real-tiles kernel)
            const unsigned b0 = (unsigned)ts * (unsigned)A.up * ESZ;
            const size_t step = (size_t)256 * A.up * ESZ;
            // DEC here means L / M: up-rate index j = (out0 + i) up + ph is kept iff M divides it, at y[j / M] (as in fir_ols.hip: tile-local
            // j below 2^20 + M, multiply-high exact for M <= 4096; n_keep = floor(n up / M) outputs exist)
            const unsigned M = (unsigned)A.dec;
            const int64_t jt = out0 * A.up + ph, q0 = DEC ? jt / A.dec : 0;
            const unsigned r0 = DEC ? (unsigned)(jt - q0 * A.dec) : 0u;
            const int64_t qleft = A.n_keep - q0;
            const int qlim = qleft > (1 << 24) ? (1 << 24) : (int)qleft;
            const unsigned jv = (unsigned)A.V * (unsigned)A.up;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                if (a < a0) continue;
                const int i = 256 * (a - a0);
                if (DEC) {
                    const unsigned j0 = r0 + (unsigned)(i + ts) * (unsigned)A.up;
                    const unsigned k0 = (unsigned)(((unsigned long long)j0 * A.dec_magic) >> 32);
                    if (REAL) {
                        const unsigned j1 = j0 + jv;
                        const unsigned k1 = (unsigned)(((unsigned long long)j1 * A.dec_magic) >> 32);
                        if (k0 * M == j0 && (int)k0 < qlim && (whole || i < lim)) A.y[q0 + k0] = v[a].x;
                        if (k1 * M == j1 && (int)k1 < qlim && (whole || i < lim - A.V)) A.y[q0 + k1] = v[a].y;
                    } else if (k0 * M == j0 && (int)k0 < qlim && (whole || i < lim)) {
                        v2d_t q;
                        q.x = v[a].x; q.y = v[a].y;
                        reinterpret_cast<v2d_t *>(A.y)[q0 + k0] = q;
                    }
                    continue;
                }
                if (REAL) {
                    if (whole || i < lim) *reinterpret_cast<double *>(ua0 + (size_t)(a - a0) * step + b0) = v[a].x;
                    if (whole || i < lim - A.V) *reinterpret_cast<double *>(ub0 + (size_t)(a - a0) * step + b0) = v[a].y;
                } else if (whole || i < lim) {
                    v2d_t q;
                    q.x = v[a].x; q.y = v[a].y;
                    *reinterpret_cast<v2d_t *>(ua0 + (size_t)(a - a0) * step + b0) = q;
                }
            }
        } else if (DEC) {
            // decimating store (multirate_FIR.dn): the tile's (the pair's) kept outputs are ONE run of y.  Every thread drops its kept samples
            // into the idle image at their output positions -- one multiply-high per sample finds them -- and the workgroup writes the run
            // with consecutive stores.  (Before: a 64-bit remainder, a 64-bit quotient and a predicated store per sample.)
            const unsigned M = (unsigned)A.dec;
            const int64_t q0 = out0 / A.dec;                 // uniform
            const unsigned r0 = (unsigned)(out0 - q0 * A.dec);
            const unsigned ob = r0 != 0 ? 1u : 0u;           // the run's first output, relative to q0
            int a0 = A.a0;   // (opaque copy: nothing of this path is hoisted out of the tile loop)
            asm volatile("" : "+s"(a0));
            __syncthreads();   // every wave has read its share of the image
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                if (a < a0) continue;
                const unsigned ga = r0 + 256u * (unsigned)(a - a0) + (unsigned)ts;
                const unsigned ka = (unsigned)(((unsigned long long)ga * A.dec_magic) >> 32);
                if (REAL) {
                    double *buf = reinterpret_cast<double *>(img);
                    const unsigned gb = ga + (unsigned)A.V;
                    const unsigned kb = (unsigned)(((unsigned long long)gb * A.dec_magic) >> 32);
                    if (ka * M == ga) buf[ka - ob] = v[a].x;
                    if (kb * M == gb) buf[kb - ob] = v[a].y;
                } else if (ka * M == ga) {
                    img[ka - ob] = v[a];
                }
            }
            __syncthreads();
            int64_t oe = q0 + (int64_t)((r0 + (REAL ? 2u : 1u) * (unsigned)A.V + M - 1) / M);
            const int64_t n_out = A.n_keep / A.dec;
            if (oe > n_out) oe = n_out;
            const int cnt = (int)(oe - (q0 + ob));
            if (REAL) {
                const double *buf = reinterpret_cast<const double *>(img);
                double *yo = A.y + q0 + ob;
                for (int i = ts; i < cnt; i += 256) __builtin_nontemporal_store(buf[i], yo + i);
            } else {
                v2d_t *yo = reinterpret_cast<v2d_t *>(A.y) + q0 + ob;
                for (int i = ts; i < cnt; i += 256) {
                    v2d_t q;
                    q.x = img[i].x; q.y = img[i].y;
                    __builtin_nontemporal_store(q, yo + i);
                }
            }
        } else if (full) {
            auto stores = [&](auto a0c) __attribute__((always_inline)) {
                constexpr int A0 = decltype(a0c)::value;
#pragma unroll
                for (int a = A0; a < 16; ++a) {
                    if (REAL) {
                        __builtin_nontemporal_store(v[a].x, (ybase + out0) + (unsigned)(256 * (a - A0) + ts));
                        __builtin_nontemporal_store(v[a].y, (ybase + out0 + A.V) + (unsigned)(256 * (a - A0) + ts));
                    } else {
                        v2d_t q;
                        q.x = v[a].x; q.y = v[a].y;
                        __builtin_nontemporal_store(q, (reinterpret_cast<v2d_t *>(ybase) + out0) + (unsigned)(256 * (a - A0) + ts));
                    }
                }
            };
            switch (A.a0) {
                case 1: stores(std::integral_constant<int, 1>{}); break;
                case 2: stores(std::integral_constant<int, 2>{}); break;
                case 3: stores(std::integral_constant<int, 3>{}); break;
                case 4: stores(std::integral_constant<int, 4>{}); break;
                case 5: stores(std::integral_constant<int, 5>{}); break;
                case 6: stores(std::integral_constant<int, 6>{}); break;
                case 7: stores(std::integral_constant<int, 7>{}); break;
                default: stores(std::integral_constant<int, 8>{}); break;
            }
        } else {
            int a0 = A.a0;   // (opaque copy: nothing of this path is hoisted out of the tile loop)
            asm volatile("" : "+s"(a0));
            if (REAL) {
                const int64_t outA = out0, outB = outA + A.V;
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    if (a < a0) continue;
                    const int64_t loc = 256 * (a - a0) + ts;
                    const int64_t ga = outA + loc, gb = outB + loc;
                    if (DEC) {
                        if (ga < A.n_keep && ga % A.dec == 0) A.y[ga / A.dec] = v[a].x;
                        if (gb < A.n_keep && gb % A.dec == 0) A.y[gb / A.dec] = v[a].y;
                    } else {
                        if (ga < A.n) __builtin_nontemporal_store(v[a].x, ybase + ga);
                        if (gb < A.n) __builtin_nontemporal_store(v[a].y, ybase + gb);
                    }
                }
            } else {
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    if (a < a0) continue;
                    const int64_t g = out0 + 256 * (a - a0) + ts;
                    v2d_t q;
                    q.x = v[a].x; q.y = v[a].y;
                    if (DEC) {
                        if (g < A.n_keep && g % A.dec == 0) reinterpret_cast<v2d_t *>(A.y)[g / A.dec] = q;
                    } else if (g < A.n) {
                        __builtin_nontemporal_store(q, reinterpret_cast<v2d_t *>(ybase) + g);
                    }
                }
            }
        }
        // a non-finite input makes every result of the tile non-finite: noted by walk step, recomputed behind the loop (careful.hpp)
        if (__builtin_expect(__any(not_finite(v[15].x) || not_finite(v[15].y)), 0)) careful_note(&ols_noted, (tile - tile_first()) / gridDim.x);
        if constexpr (PREF) {
            have = pre;
            if (pre) {
#pragma unroll
                for (int b = 0; b < 16; ++b) v[b] = nx[b];
            }
        }
        // The image must be free for the next tile -- but in the plain complex tile the last inverse pass of thread (hi4, lo4) reads img[k1][hi4][lo4], k1 = 0 .. 15, and
        // pass 1 of the same thread overwrites exactly those sixteen elements: between the barrier in front of that read and the one behind pass 1 a thread meets only
        // its own elements, in program order (round 6; fir_ols.hip's tile dropped the same barrier).  The decimating store gathers in the image and keeps it, and so
        // do the forms that share this loop with it.
        if constexpr (REAL || DEC || UP || XR) __syncthreads();
    }
    const unsigned long long noted = careful_noted(&ols_noted);
    if (__builtin_expect(noted != 0, 0)) {
        int64_t k = 0;
        for (int64_t w = tile_first(); w < A.ntiles; w += gridDim.x, ++k) {
            if (!careful_step_noted(noted, k)) continue;
            const int64_t tin = UP ? (int64_t)((unsigned)w / (unsigned)A.up) : w;
            const int ph = UP ? (int)((unsigned)w % (unsigned)A.up) : 0;
            OlsCareful c;
            c.x = A.x; c.y = A.y; c.n = A.n; c.n_hist = A.n_hist; c.n_keep = A.n_keep; c.up_pitch = A.up_pitch;
            c.V = A.V; c.dec = A.dec; c.up = A.up;
            c.L = XR ? 2 * A.up : A.up; c.p0 = XR ? 2 * ph : ph;
            c.cf = A.cf;
            careful_ols_tile<double, REAL, DEC, UP, XR>(c, tin, ph, t);
        }
    }
}

}  // namespace

struct Ols64Plan {
    int ov = 0, V = 0;
    cdd *Hp = nullptr, *W1 = nullptr, *W2 = nullptr;
};

void fir_ols64_free(Ols64Plan *p)
{
    if (!p) return;
    if (p->Hp) (void)hipFree(p->Hp);
    if (p->W1) (void)hipFree(p->W1);
    if (p->W2) (void)hipFree(p->W2);
    delete p;
}

bool fir_ols64_supported(const FirHandle *h)
{
    if (h->ntaps < 2 || h->ntaps - 1 > 2048) return false;
    return h->dtype == SKDSP_C128 || (h->dtype == SKDSP_F64 && !h->taps_complex);
}

// Tables of one plan: `up` phase filters (phase q: taps up * b[q + up t]) as `up` consecutive H tables; up = 1: the filter itself.
static int build_plan64(const FirHandle *h, int up, Ols64Plan **out, bool paired = false)
{
    typedef std::complex<long double> cl;
    const int T = (h->ntaps + up - 1) / up;
    const int ntab = paired ? up / 2 : up;   // paired (real taps, even up): table k holds phase 2k + i phase 2k+1
    Ols64Plan *p = new Ols64Plan();
    p->ov = ((T - 1 + 255) / 256) * 256;
    if (p->ov == 0) p->ov = 256;
    p->V = kN64 - p->ov;
    // H = DFT_4096(b) / N in long double (plain O(N P) sums: once per handle)
    const int comp = h->taps_complex ? 2 : 1;
    const long double two_pi = 6.283185307179586476925286766559L;
    std::vector<cl> wn(kN64);
    for (int k = 0; k < kN64; ++k) wn[k] = cl(cosl(two_pi * k / kN64), -sinl(two_pi * k / kN64));
    std::vector<double> Hp((size_t)2 * kN64 * ntab), W1(2 * 256), W2(2 * 16);
    // each table: the 4096-point DFT of the phase's taps, radix-2 in long double (the O(N T) sums this replaces took seconds for the L tables
    // of a long interpolator)
    std::vector<cl> f(kN64);
    for (int q = 0; q < ntab; ++q) {
        std::fill(f.begin(), f.end(), cl(0, 0));
        for (int t = 0; t < T; ++t) {
            const int j = (paired ? 2 * q : q) + up * t;
            if (j >= h->ntaps) break;
            f[t] = paired ? cl(h->taps_host[j], j + 1 < h->ntaps ? h->taps_host[j + 1] : 0.0)
                          : (comp == 2 ? cl(h->taps_host[2 * j], h->taps_host[2 * j + 1]) : cl(h->taps_host[j], 0));
        }
        for (int i = 1, j = 0; i < kN64; ++i) {   // bit reversal
            int bit = kN64 >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(f[i], f[j]);
        }
        for (int len = 2; len <= kN64; len <<= 1)
            for (int i = 0; i < kN64; i += len)
                for (int k = 0; k < len / 2; ++k) {
                    const cl u = f[i + k], w = f[i + k + len / 2] * wn[(size_t)k * (kN64 / len)];
                    f[i + k] = u + w;
                    f[i + k + len / 2] = u - w;
                }
        for (int k = 0; k < kN64; ++k) {
            cl acc = f[k];
            acc *= (long double)up / (long double)kN64;   // (up = 1: 1 / N; else the gain L of multirate_FIR.up as well)
            const int k1 = k & 15, k2 = (k >> 4) & 15, k3 = k >> 8;
            const size_t idx = (size_t)q * kN64 + (size_t)k3 * 256 + 16 * k1 + k2;
            Hp[2 * idx] = (double)acc.real();
            Hp[2 * idx + 1] = (double)acc.imag();
        }
    }
    for (int t = 0; t < 256; ++t) { W1[2 * t] = (double)wn[t].real(); W1[2 * t + 1] = (double)wn[t].imag(); }
    for (int c = 0; c < 16; ++c) { W2[2 * c] = (double)wn[16 * c].real(); W2[2 * c + 1] = (double)wn[16 * c].imag(); }
    hipError_t e;
    if ((e = hipMalloc((void **)&p->Hp, Hp.size() * 8)) != hipSuccess || (e = hipMalloc((void **)&p->W1, W1.size() * 8)) != hipSuccess ||
        (e = hipMalloc((void **)&p->W2, W2.size() * 8)) != hipSuccess ||
        (e = hipMemcpy(p->Hp, Hp.data(), Hp.size() * 8, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->W1, W1.data(), W1.size() * 8, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->W2, W2.data(), W2.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) {
        fir_ols64_free(p);
        return hip_fail(e, "ols64 tables", __FILE__, __LINE__);
    }
    *out = p;
    return SKDSP_OK;
}

static int ensure_plan64(FirHandle *h)
{
    if (h->ols64) return SKDSP_OK;
    return build_plan64(h, 1, &h->ols64);
}

int fir_ols64_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, void *y, hipStream_t s, int dec)
{
    note_path("fir_ols64");
    if (n <= 0) return SKDSP_OK;
    if (dec > 1) n = (n / dec) * dec;
    if (n <= 0) return SKDSP_OK;
    SK_CHECK(fir_ols64_supported(h), SKDSP_ERR_UNSUPPORTED, "fir_ols64: needs complex128 (or float64 with real taps) and 2..2049 taps");
    int rc = ensure_plan64(h);
    if (rc) return rc;
    Ols64Plan *p = h->ols64;
    Ols64Args A;
    A.x = (const double *)x; A.y = (double *)y; A.n = n; A.n_hist = n_hist;
    A.Hp = p->Hp; A.W1 = p->W1; A.W2 = p->W2;
    A.ov = p->ov; A.V = p->V; A.a0 = p->ov / 256;
    const bool real = h->dtype == SKDSP_F64;
    int64_t ntiles = (n + p->V - 1) / p->V;
    if (real) ntiles = (ntiles + 1) / 2;
    A.ntiles = ntiles;
    A.dec = dec > 1 ? dec : 1;
    A.dec_magic = A.dec > 1 ? (unsigned)((((unsigned long long)1 << 32) + A.dec - 1) / A.dec) : 0u;
    A.n_keep = n;
    A.up = 1; A.up_pitch = 0;
    if ((rc = fir_careful(h, &A.cf))) return rc;
    int64_t grid = 2 * (int64_t)ctx().num_cus;
    if (grid > ntiles) grid = ntiles;
    if (A.dec > 1) {
        if (real) hipLaunchKernelGGL((ols64_tile_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((ols64_tile_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
    } else {
        if (real) hipLaunchKernelGGL((ols64_tile_kernel<true, false>), dim3((unsigned)grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((ols64_tile_kernel<false, false>), dim3((unsigned)grid), dim3(256), 0, s, A);
    }
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

// multirate_FIR.up with long phases (see fir_ols.hip, fir_ols_up_launch): complex128, float64 with real taps; 2..2049 taps per phase
bool fir_ols64_up_supported(const FirHandle *h, int L)
{
    if (L < 2 || L > 256) return false;   // (the every-M-th store: L <= 64, checked at launch)
    const int T = (h->ntaps + L - 1) / L;
    if (T < 2 || T - 1 > 2048) return false;
    return h->dtype == SKDSP_C128 || (h->dtype == SKDSP_F64 && !h->taps_complex);
}

// float64 signals, real taps, even L, no decimation, a 16-byte aligned destination: the phases run in pairs through the complex tile
bool fir_ols64_up_pairs(const FirHandle *h, int L, int dec, const void *y)
{
    return opt().fir_up_pair && h->dtype == SKDSP_F64 && !h->taps_complex && L % 2 == 0 && dec <= 1 && ((uintptr_t)y & 15) == 0;
}

int fir_ols64_up_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, int L, void *y, hipStream_t s, int dec, int64_t rows_pitch, int paired_in)
{
    note_path("fir_ols64_up");
    if (n <= 0) return SKDSP_OK;
    SK_CHECK(dec >= 1 && dec <= 4096 && (dec == 1 || L <= 64), SKDSP_ERR_UNSUPPORTED, "fir_ols64_up: L / M = %d / %d (the fused L / M store takes L <= 64, M <= 4096)", L, dec);
    SK_CHECK(fir_ols64_up_supported(h, L), SKDSP_ERR_UNSUPPORTED, "fir_ols64_up: needs complex128 (or float64 with real taps), 2 <= L <= 256, 2..2049 taps per phase");
    const bool paired = paired_in != 0;
    SK_CHECK(!paired || fir_ols64_up_pairs(h, L, dec, y), SKDSP_ERR_BADARG, "fir_ols64_up: phases in pairs need float64, real taps, an even L, no decimation and a 16-byte aligned destination");
    const int key = paired ? -L : L;
    Ols64Plan *p = nullptr;
    for (auto &u : h->ols64_up)
        if (u.L == key) p = u.plan;
    if (!p) {
        int rc = build_plan64(h, L, &p, paired);
        if (rc) return rc;
        h->ols64_up.push_back(FirHandle::Ols64Up{key, p});
    }
    Ols64Args A;
    A.x = (const double *)x; A.y = (double *)y; A.n = n; A.n_hist = n_hist;
    A.Hp = p->Hp; A.W1 = p->W1; A.W2 = p->W2;
    A.ov = p->ov; A.V = p->V; A.a0 = p->ov / 256;
    const bool real = h->dtype == SKDSP_F64 && !paired;
    const int phases = paired ? L / 2 : L;
    int64_t ntiles = (n + p->V - 1) / p->V;
    if (real) ntiles = (ntiles + 1) / 2;
    ntiles *= phases;
    SK_CHECK(ntiles < (int64_t)1 << 31, SKDSP_ERR_BADARG, "fir_ols64_up: too many tiles");
    A.ntiles = ntiles;
    A.dec = dec;
    A.dec_magic = dec > 1 ? (unsigned)((((unsigned long long)1 << 32) + dec - 1) / dec) : 0u;
    A.n_keep = dec > 1 ? (n * L) / dec : n;   // (L / M: the number of outputs)
    A.up = phases;
    A.up_pitch = dec > 1 ? 0 : rows_pitch;
    {
        int rc = fir_careful(h, &A.cf);
        if (rc) return rc;
    }
    int64_t grid = 2 * (int64_t)ctx().num_cus;
    if (grid > ntiles) grid = ntiles;
    if (paired) {
        if (phases == 1 && A.up_pitch == 0) A.up_pitch = 1;   // (L = 2 is one pair: "row 0" is the output, written with the plain filter's stores)
        hipLaunchKernelGGL((ols64_tile_kernel<false, false, true, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
    } else if (dec > 1) {
        if (real) hipLaunchKernelGGL((ols64_tile_kernel<true, true, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((ols64_tile_kernel<false, true, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
    } else {
        if (real) hipLaunchKernelGGL((ols64_tile_kernel<true, false, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((ols64_tile_kernel<false, false, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
    }
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

}  // namespace skdsp
