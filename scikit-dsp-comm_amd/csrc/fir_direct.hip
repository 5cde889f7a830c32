// fir_direct.hip -- direct-form / polyphase FIR for gfx950 (MI355X).
//
// One kernel family serves every time-domain FIR entry of the path:
//   .filter  y[n] = sum_k b[k] x[n-k]                      multirate_helper.py:104-109
//   .up      y    = lfilter(b,[1], L*upsample(x,L))         multirate_helper.py:112-118
//   .dn      y    = downsample(lfilter(b,[1],x), M)         multirate_helper.py:121-127
//   updn     y    = downsample(.up(x,L), M)                 (BASELINE.json config 3)
// through the polyphase identity: with j = m*M,
//   y[m] = L * sum_t b[(j mod L) + L t] * x[(j div L) - t]
// so zero-stuffed samples are never materialised and discarded outputs are never
// computed.  Outputs are grouped in classes c = m mod L' (L' = L/gcd(L,M)): within a
// class the tap phase is fixed and the input index advances by q = M/gcd(L,M) per
// output, so a whole wave uses ONE tap per step (scalar load, SGPR operand) and reads
// a stride-q run of the LDS-staged input window.
//
// Layout: a 256-thread workgroup stages the input window
//   [q*s0 - (T-1), q*(s0 + s_tile) + q)   (T = ceil(P/L) taps per phase)
// once into LDS with coalesced loads (zero-filled outside [-n_hist, n)), then loops
// over the L' classes; each thread keeps R accumulators (outputs s = s0 + tid + 256 r).
// Accumulation is in the signal precision (f32 / f64), taps applied in k order.
//
// Roofline note: this kernel is FP32-VALU/LDS bound for long filters (4*P flop per
// c64 sample); long .filter calls go to fir_ols.hip instead.  It is the HBM-bound
// choice only for short filters.
#include "skdsp_internal.hpp"
#include <type_traits>
#include <cstring>
#include <numeric>
#include <cstdlib>

namespace skdsp {

// ------------------------------------------------------------------ arithmetic
__device__ inline void mac(float &a, float b, float x) { a = fmaf(b, x, a); }
__device__ inline void mac(double &a, double b, double x) { a = fma(b, x, a); }
__device__ inline void mac(float2 &a, float b, float2 x) { a.x = fmaf(b, x.x, a.x); a.y = fmaf(b, x.y, a.y); }
__device__ inline void mac(double2 &a, double b, double2 x) { a.x = fma(b, x.x, a.x); a.y = fma(b, x.y, a.y); }
__device__ inline void mac(float2 &a, float2 b, float2 x)
{
    a.x = fmaf(b.x, x.x, a.x); a.x = fmaf(-b.y, x.y, a.x);
    a.y = fmaf(b.x, x.y, a.y); a.y = fmaf(b.y, x.x, a.y);
}
__device__ inline void mac(double2 &a, double2 b, double2 x)
{
    a.x = fma(b.x, x.x, a.x); a.x = fma(-b.y, x.y, a.x);
    a.y = fma(b.x, x.y, a.y); a.y = fma(b.y, x.x, a.y);
}
template <typename X> __device__ inline X zero_of();
template <> __device__ inline float zero_of<float>() { return 0.f; }
template <> __device__ inline double zero_of<double>() { return 0.; }
template <> __device__ inline float2 zero_of<float2>() { return make_float2(0.f, 0.f); }
template <> __device__ inline double2 zero_of<double2>() { return make_double2(0., 0.); }
__device__ inline float add_of(float a, float b) { return a + b; }
__device__ inline double add_of(double a, double b) { return a + b; }
__device__ inline float2 add_of(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ inline double2 add_of(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ inline float scl(float a, float s) { return a * s; }
__device__ inline double scl(double a, double s) { return a * s; }
__device__ inline float2 scl(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ inline double2 scl(double2 a, double s) { return make_double2(a.x * s, a.y * s); }

template <typename X> struct ScalarOf { using type = float; };
template <> struct ScalarOf<double> { using type = double; };
template <> struct ScalarOf<double2> { using type = double; };

struct PolyArgs {
    int64_t n;        // input samples
    int64_t n_hist;   // valid samples before x[0]
    int64_t n_out;    // total outputs
    int64_t n_s;      // outputs per class (max over classes)
    int T;            // taps per phase
    int L, M;
    int Lp;           // classes  L' = L / gcd(L,M)
    int q;            // input stride per output within a class
    int s_tile;       // outputs per class per workgroup (<= 256*R)
    int win;          // staged window length in samples
    CarefulFir cf;    // for the re-evaluation of non-finite results (careful.hpp)
};

template <typename X, typename B, int R>
__global__ __launch_bounds__(256) void fir_poly_kernel(const X *__restrict__ x, const B *__restrict__ bank, PolyArgs a,
                                                       X *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int poly_nf;   // some result of this workgroup came out non-finite (see the end of the kernel)
    if (threadIdx.x == 0) poly_nf = 0;
    X *win = reinterpret_cast<X *>(smem_raw);
    using S = typename ScalarOf<X>::type;

    const int tid = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * a.s_tile;
    const int64_t w0 = (int64_t)a.q * s0 - (a.T - 1);  // global input index of win[0]

    // ---- stage the window (coalesced; zero outside [-n_hist, n)) ----
    // eight loads in flight per thread: a load-wait-store loop costs one HBM round trip per 256 samples
    for (int i0 = tid; i0 < a.win; i0 += 256 * 8) {
        X v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 256 * u;
            const int64_t g = w0 + i;
            v[u] = zero_of<X>();
            if (i < a.win && g >= -a.n_hist && g < a.n) v[u] = x[g];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 256 * u;
            if (i < a.win) win[i] = v[u];
        }
    }
    __syncthreads();

    const S gain = (S)a.L;
    for (int c = 0; c < a.Lp; ++c) {
        const int64_t cm = (int64_t)c * a.M;
        const int phi = (int)(cm % a.L);
        const int ic = (int)(cm / a.L);
        const B *__restrict__ bp = bank + (size_t)phi * a.T;
        X acc[R];
        int off[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] = zero_of<X>();
            // window offset of x[i_c + q*s] for this thread's r-th output, tap 0
            off[r] = ic + a.q * (tid + 256 * r) + (a.T - 1);
            if (tid + 256 * r >= a.s_tile) off[r] = a.T - 1;  // idle slot: stay in range
        }
        // Blocked summation: taps are accumulated 16 at a time into a fresh partial sum that
        // is then folded into the running total.  Rounding error grows like
        // sqrt(16) + sqrt(T/16) instead of sqrt(T) ulps, which keeps 1024-tap float32
        // filters inside the 1e-6 parity bound at the cost of one add per 16 taps.
        int t = 0;
        for (; t + 16 <= a.T; t += 16) {
            X part[R];
#pragma unroll
            for (int r = 0; r < R; ++r) part[r] = zero_of<X>();
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                const B b0 = bp[t + u], b1 = bp[t + u + 1], b2 = bp[t + u + 2], b3 = bp[t + u + 3];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const X *p = win + off[r] - (t + u);
                    mac(part[r], b0, p[0]);
                    mac(part[r], b1, p[-1]);
                    mac(part[r], b2, p[-2]);
                    mac(part[r], b3, p[-3]);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = add_of(acc[r], part[r]);
        }
        if (t < a.T) {
            X part[R];
#pragma unroll
            for (int r = 0; r < R; ++r) part[r] = zero_of<X>();
            for (; t < a.T; ++t) {
                const B b0 = bp[t];
#pragma unroll
                for (int r = 0; r < R; ++r) mac(part[r], b0, win[off[r] - t]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = add_of(acc[r], part[r]);
        }
        {   // a non-finite result: did a real tap meet the sample, or only the zero padding of the phase table?  Looked at behind the loop (careful.hpp)
            bool bad = false;
#pragma unroll
            for (int r = 0; r < R; ++r) bad |= not_finite(acc[r]);
            if (__builtin_expect(__any(bad), 0)) poly_nf = 1;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int sl = tid + 256 * r;
            const int64_t m = (int64_t)c + (int64_t)a.Lp * (s0 + sl);
            if (sl < a.s_tile && m < a.n_out) y[m] = (a.L == 1) ? acc[r] : scl(acc[r], gain);
        }
    }
    // non-finite results among this workgroup's outputs m in [Lp s0, Lp (s0 + s_tile)): re-evaluated by the reference's sum (careful.hpp)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (__builtin_expect(*reinterpret_cast<volatile int *>(&poly_nf) != 0, 0))
        careful_fir_recheck<typename CarefulOf<X>::T, CarefulOf<X>::CX>(x, y, a.n_hist, a.n_out, (int64_t)a.Lp * s0, (int64_t)a.Lp * a.s_tile, a.L, a.M, a.cf, tid);
}

// ------------------------------------------------------------------ sliding window
// Register sliding-window variant: a thread owns R CONSECUTIVE outputs of a class, so the
// taps t = q m + j of one residue j form a stride-1 FIR over S_j(n) = x[i_c + q(s+n) - j]:
// per tap ONE new LDS element feeds R FMAs (the generic kernel above needs R reads).
//
// The staged window is cut into groups of P = q R samples stored with pitch P+1 (odd
// lane stride => conflict-free b32/b64 reads).  Thread tid's R outputs start at group
// G + tid; the R window elements a block of R taps needs sit in ONE group:
//     block beta of (class c, residue j):  group G + tid - beta - 1, offsets rho + q i
// with rho = i_c - j + q delta in [0, q) (delta in {0,1} prepends one zero tap when
// i_c < j).  So the inner loop is: R ds_reads at constant offsets from a pointer that
// steps down by one group, R scalar taps from a zero-padded per-(c, j) table, R*R FMAs.
// The window lives in registers as wr[R] (current) + nw[R] (next), statically indexed.
// Summation: R taps -> part, 16 parts -> mid, mid -> acc (keeps 4097-tap float32 sums
// inside the 1e-6 parity bound).
struct SwArgs {
    int64_t n, n_hist, n_out;
    int L, Lp, q;
    int G;      // groups staged to the left of the first output group
    int nB;     // tap blocks per (class, residue) = table pitch / R
    int win;    // logical window length = (G + 256) * q * R
    int tap_off;  // byte offset of the two per-class tap buffers in LDS (-1: none, taps come through scalar loads)
    int tap_cnt;  // floats per class = q * nB * R
    int half_last;  // taps 4..7 of the last block of every (class, residue) row are zero padding
    int out_tile;  // 1: out_off is ONE image of all Lp classes of the workgroup's slots (256*R*Lp elements)
    int out_off;  // byte offset of the four wave-private output transposition tiles (-1: store directly)
    int M;        // of the call, for the re-evaluation of non-finite results (careful.hpp)
    CarefulFir cf;
};


// ---- complex64 x real taps, R = 8: packed FP32 with the tap broadcast done by op_sel ----------
// v_pk_fma_f32 multiplies (re, im) by (tap, tap); hipcc builds that pair with two v_mov per tap and
// rotates the register window with 16 more per block (46 v_mov per 64 useful v_pk_fma).  Here one
// 64-bit register pair holds two consecutive taps and op_sel picks the half, and the window ping-pongs
// between two register sets, so a block is 64 v_pk_fma/mul + 8 v_pk_add.  One asm statement covers
// a tap pair x 8 outputs (hipcc pads every asm boundary with s_nop).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef v2f tap2_t;  // two consecutive taps in one 64-bit VGPR pair

// p[r] (+)= t_lo * w[r+1] + t_hi * w[r],  r = 0..7  (w = 9 consecutive window registers, oldest first)
__device__ __forceinline__ void pair_first(v2f (&p)[8], const v2f w0, const v2f w1, const v2f w2, const v2f w3, const v2f w4,
                                           const v2f w5, const v2f w6, const v2f w7, const v2f w8, const tap2_t tt)
{
    asm("v_pk_mul_f32 %0, %9,  %16 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %10, %16 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %2, %11, %16 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %3, %12, %16 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %4, %13, %16 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %5, %14, %16 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %6, %15, %16 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %7, %17, %16 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %8,  %16, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %1, %9,  %16, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %2, %10, %16, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %3, %11, %16, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %4, %12, %16, %4 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %5, %13, %16, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %6, %14, %16, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %7, %15, %16, %7 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7])
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7), "v"(tt), "v"(w8));
}

__device__ __forceinline__ void pair_next(v2f (&p)[8], const v2f w0, const v2f w1, const v2f w2, const v2f w3, const v2f w4,
                                          const v2f w5, const v2f w6, const v2f w7, const v2f w8, const tap2_t tt)
{
    asm("v_pk_fma_f32 %0, %9,  %16, %0 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %10, %16, %1 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %2, %11, %16, %2 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %3, %12, %16, %3 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %4, %13, %16, %4 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %5, %14, %16, %5 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %6, %15, %16, %6 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %7, %17, %16, %7 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %8,  %16, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %1, %9,  %16, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %2, %10, %16, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %3, %11, %16, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %4, %12, %16, %4 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %5, %13, %16, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %6, %14, %16, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %7, %15, %16, %7 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7), "v"(tt), "v"(w8));
}

// one block of 8 taps (4 SGPR pairs) over the window {nxt[0..7] (older), cur[0..7]}: tap u feeds
// output r from W[8 + r - u]
// `half`: only taps 0..3 of the block are non-zero (the zero-padded tail of a tap table) -- uniform
__device__ __forceinline__ void sw_block8(v2f (&mid)[8], const v2f (&cur)[8], const v2f (&nxt)[8], const tap2_t (&tt)[4],
                                          const bool half)
{
    v2f p[8];
    pair_first(p, nxt[7], cur[0], cur[1], cur[2], cur[3], cur[4], cur[5], cur[6], cur[7], tt[0]);
    pair_next(p, nxt[5], nxt[6], nxt[7], cur[0], cur[1], cur[2], cur[3], cur[4], cur[5], tt[1]);
    if (!half) {
        pair_next(p, nxt[3], nxt[4], nxt[5], nxt[6], nxt[7], cur[0], cur[1], cur[2], cur[3], tt[2]);
        pair_next(p, nxt[1], nxt[2], nxt[3], nxt[4], nxt[5], nxt[6], nxt[7], cur[0], cur[1], tt[3]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) mid[r] += p[r];
}

// LPT > 0 (= Lp, complex64 x real taps only): the class loop is unrolled and all Lp classes of a
// thread's slots stay in registers, so that the epilogue can write the interleaved output as full
// 512-byte rows (per-class stores fill a quarter of every line at a time; L2 merges them, but at
// the price of 0.15 ms of config 3's 0.97).
template <typename X, typename B, int R, int Q, int LPT = 0>
__global__ __launch_bounds__(256) void fir_sw_kernel(const X *__restrict__ x, const B *__restrict__ taps,
                                                     const int *__restrict__ rho_tab, SwArgs a, X *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int sw_nf;   // some result of this workgroup came out non-finite (see recheck_outputs)
    if (threadIdx.x == 0) sw_nf = 0;
    X *win = reinterpret_cast<X *>(smem_raw);
    using S = typename ScalarOf<X>::type;
    const int tid = threadIdx.x;
    const int q = Q > 0 ? Q : a.q;
    const int P = q * R;
    const int64_t s0 = (int64_t)blockIdx.x * (256 * R);
    const int64_t w0 = (int64_t)q * s0 - (int64_t)P * a.G;

    // ---- stage the window.  Interior workgroups (the whole window inside [-n_hist, n), 16-byte
    // aligned) take 16-byte loads with no per-sample bounds logic; eight loads are in flight per
    // thread either way (a load-wait-store loop costs one HBM round trip per 256 samples).
    constexpr int VEC = 16 / (int)sizeof(X);
    const X *src = x + w0;
    const bool interior = w0 >= -a.n_hist && w0 + a.win <= a.n && (P % VEC) == 0 && (a.win % VEC) == 0 &&
                          (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    if (interior) {
        const int nv = a.win / VEC;
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        for (int k0 = tid; k0 < nv; k0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 256 * u;
                if (k < nv) v[u] = src4[k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 256 * u;
                if (k < nv) {
                    const int i = k * VEC;
                    X *dst = win + i + i / P;  // the VEC samples share a group (P % VEC == 0)
                    const X *e = reinterpret_cast<const X *>(&v[u]);
#pragma unroll
                    for (int t = 0; t < VEC; ++t) dst[t] = e[t];
                }
            }
        }
    } else {
        for (int i0 = tid; i0 < a.win; i0 += 256 * 8) {
            X v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u;
                const int64_t g = w0 + i;
                v[u] = zero_of<X>();
                if (i < a.win && g >= -a.n_hist && g < a.n) v[u] = x[g];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u;
                if (i < a.win) win[i + i / P] = v[u];
            }
        }
    }
    // LDS copy of ONE class's tap table at a time, double-buffered (complex64 x real taps path):
    // class c+1's taps are staged while class c is computed; the barrier on top of every class
    // publishes them and retires the buffer they replace.  (The whole table would push the
    // workgroup past a third of the CU's LDS for the config-3 shape.)
    auto stage_taps = [&](int c) {
        float4 *dst = reinterpret_cast<float4 *>(smem_raw + a.tap_off + (c & 1) * a.tap_cnt * 4);
        const float4 *src = reinterpret_cast<const float4 *>(taps + (size_t)c * a.tap_cnt);
        for (int i = tid; i < a.tap_cnt / 4; i += 256) dst[i] = src[i];
    };
    if (a.tap_off >= 0) stage_taps(0);
    __syncthreads();

    const S gain = (S)a.L;
    auto class_body = [&](const int c, X (&acc)[R]) __attribute__((always_inline)) {
        if (a.tap_off >= 0) {
            if (c > 0) __syncthreads();
            if (c + 1 < a.Lp) stage_taps(c + 1);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = zero_of<X>();
        for (int j = 0; j < q; ++j) {
            const int cj = c * q + j;
            const int rho = rho_tab[cj];
            if (rho < 0) continue;  // residue j has no taps
            const B *__restrict__ tb = taps + (size_t)cj * a.nB * R;
            const X *grp = win + (size_t)(a.G + tid) * (P + 1) + rho;
            if constexpr (std::is_same<X, float2>::value && std::is_same<B, float>::value && R == 8) if (a.tap_off >= 0) {
                const v2f *g2 = reinterpret_cast<const v2f *>(grp);
                const tap2_t *tq = reinterpret_cast<const tap2_t *>(smem_raw + a.tap_off + (c & 1) * a.tap_cnt * 4) + (size_t)j * a.nB * 4;
                v2f wa[8], wb[8], wc[8], mid2[8];
                v2f *acc2 = reinterpret_cast<v2f *>(acc);  // the class accumulator itself (X = float2)
                tap2_t ta[4], tb2[4], tc[4];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    wa[e] = g2[q * e];
                    wb[e] = g2[q * e - (P + 1)];
                    mid2[e] = v2f{0.f, 0.f};
                }
#pragma unroll
                for (int u_ = 0; u_ < 4; ++u_) ta[u_] = tq[u_];
                // software pipeline, three register windows: while block beta runs on (W_beta, W_beta+1),
                // the LDS reads of W_beta+2 and of block beta+1's taps are in flight.  The taps come from
                // an LDS copy of the table, not from scalar loads: lgkmcnt counts SMEM and LDS together and
                // SMEM returns out of order, so any pending s_load forces a full drain before the math.
                int beta = 0;
#define SK_SW_STEP(CUR, NXT, PRE, TCUR, TPRE)                                                        \
    {                                                                                                \
        /* what this block consumes was requested one block ago: have hipcc wait for it HERE, before */ \
        /* the next requests go out, instead of draining those too right before the math            */ \
        asm volatile("" ::"v"(TCUR[0]), "v"(TCUR[1]), "v"(TCUR[2]), "v"(TCUR[3]), "v"(NXT[0]), "v"(NXT[1]), "v"(NXT[2]),   \
                     "v"(NXT[3]), "v"(NXT[4]), "v"(NXT[5]), "v"(NXT[6]), "v"(NXT[7]));                \
        const int kw = beta + 2 <= a.nB ? beta + 2 : a.nB;                                           \
        const v2f *gp = g2 - (size_t)kw * (P + 1);                                                   \
        const tap2_t *tqn = tq + (size_t)(beta + 1 < a.nB ? beta + 1 : beta) * 4;                    \
        _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) TPRE[u_] = tqn[u_];                         \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) PRE[i] = gp[q * i];                            \
        sw_block8(mid2, CUR, NXT, TCUR, a.half_last && beta == a.nB - 1);                            \
        ++beta;                                                                                      \
        if ((beta & 15) == 0) {                                                                      \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                          \
                acc2[r] += mid2[r];                                                                  \
                mid2[r] = v2f{0.f, 0.f};                                                             \
            }                                                                                        \
        }                                                                                            \
        if (beta >= a.nB) break;                                                                     \
    }
                for (;;) {
                    SK_SW_STEP(wa, wb, wc, ta, tb2)
                    SK_SW_STEP(wb, wc, wa, tb2, tc)
                    SK_SW_STEP(wc, wa, wb, tc, ta)
                }
#undef SK_SW_STEP
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    acc2[r] += mid2[r];
                }
                continue;
            }
            X wr[R];
#pragma unroll
            for (int e = 0; e < R; ++e) wr[e] = grp[q * e];
            X mid[R];
#pragma unroll
            for (int r = 0; r < R; ++r) mid[r] = zero_of<X>();
            // taps are wave-uniform (scalar loads); request block beta+1 while block beta is
            // being applied -- a load-then-wait per block left every wave stalled ~85 % of the time
            B tcur[R];
#pragma unroll
            for (int u_ = 0; u_ < R; ++u_) tcur[u_] = tb[u_];
            for (int beta = 0; beta < a.nB; ++beta) {
                grp -= (P + 1);
                B tnext[R];
                const B *__restrict__ tbn = tb + (size_t)(beta + 1 < a.nB ? beta + 1 : beta) * R;
#pragma unroll
                for (int u_ = 0; u_ < R; ++u_) tnext[u_] = tbn[u_];
                X nw[R];
#pragma unroll
                for (int i = 0; i < R; ++i) nw[i] = grp[q * i];
                X part[R];
#pragma unroll
                for (int r = 0; r < R; ++r) part[r] = zero_of<X>();
#pragma unroll
                for (int u_ = 0; u_ < R; ++u_) {
                    const B b = tcur[u_];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int e = r - u_;
                        mac(part[r], b, e >= 0 ? wr[e >= 0 ? e : 0] : nw[e < 0 ? e + R : 0]);
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    mid[r] = add_of(mid[r], part[r]);
                    wr[r] = nw[r];
                }
#pragma unroll
                for (int u_ = 0; u_ < R; ++u_) tcur[u_] = tnext[u_];
                if ((beta & 15) == 15) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[r] = add_of(acc[r], mid[r]);
                        mid[r] = zero_of<X>();
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = add_of(acc[r], mid[r]);
        }
    };

    // a non-finite result: did a real tap meet the sample, or only the zero padding of the tap blocks?  Looked at behind the stores (careful.hpp)
    auto recheck = [&](X (&acc)[R]) __attribute__((always_inline)) {
        bool bad = false;
#pragma unroll
        for (int r = 0; r < R; ++r) bad |= not_finite(acc[r]);
        if (__builtin_expect(__any(bad), 0)) sw_nf = 1;
    };
    // ... of this workgroup's outputs m in [Lp s0, Lp (s0 + 256 R))
    auto recheck_outputs = [&](int lp) __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (__builtin_expect(*reinterpret_cast<volatile int *>(&sw_nf) != 0, 0))
            careful_fir_recheck<typename CarefulOf<X>::T, CarefulOf<X>::CX>(x, y, a.n_hist, a.n_out, (int64_t)lp * s0, (int64_t)lp * 256 * R, a.L, a.M, a.cf, tid);
    };
    if constexpr (LPT > 0) {
        X accs[LPT][R];
#pragma unroll
        for (int c = 0; c < LPT; ++c) {
            class_body(c, accs[c]);
            recheck(accs[c]);
        }
        // element e of a wave's output run = LPT * slot + class; 16 lanes (16*LPT*R consecutive
        // elements) go through the wave-private tile per pass and leave as 512-byte rows
        static_assert(16 * (LPT * R + 1) <= 64 * (R + 1), "output tile too small");
        __syncthreads();  // the tiles reuse the front of the window image: every wave is done with it
        const int wave = tid >> 6, lane = tid & 63;
        X *ot = reinterpret_cast<X *>(smem_raw + a.out_off) + (size_t)wave * (64 * (R + 1));
        const int64_t mw = (int64_t)LPT * (s0 + (int64_t)R * 64 * wave);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if ((lane >> 4) == p) {
                X *row = ot + (lane & 15) * (LPT * R + 1);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int c = 0; c < LPT; ++c) row[LPT * r + c] = scl(accs[c][r], gain);  // LPT > 1 means L > 1
            }
#pragma unroll
            for (int k = 0; k < LPT * R / 4; ++k) {
                const int e = lane + 64 * k;
                const X v = ot[e + e / (LPT * R)];
                const int64_t m = mw + (int64_t)p * (16 * LPT * R) + e;
                if (m < a.n_out) y[m] = v;
            }
        }
        recheck_outputs(LPT);
        return;
    }
    for (int c = 0; c < a.Lp; ++c) {
        X acc[R];
        class_body(c, acc);
        recheck(acc);
        const int64_t sb = s0 + (int64_t)R * tid;
        if (a.out_tile) {
            // every class of the workgroup's slots is parked in one LDS image laid out like the
            // output run itself (slot-major, class-minor); it leaves as full rows after the loop
            X *tile = reinterpret_cast<X *>(smem_raw + a.out_off);
#pragma unroll
            for (int r = 0; r < R; ++r) tile[(size_t)(R * tid + r) * a.Lp + c] = scl(acc[r], gain);
        } else if (a.out_off >= 0) {
            // a thread's R outputs of a class sit Lp*R elements apart from its neighbour's, so a direct
            // store touches 64 lines per instruction.  Transpose through a wave-private LDS tile
            // (pitch R+1) so that consecutive lanes hold consecutive slots: 64*Lp elements per
            // instruction instead (config 3: 16 lines x 32 B, and 0.25 ms of 0.97 back).
            const int wave = tid >> 6, lane = tid & 63;
            X *ot = reinterpret_cast<X *>(smem_raw + a.out_off) + (size_t)wave * (64 * (R + 1));
#pragma unroll
            for (int r = 0; r < R; ++r) ot[lane * (R + 1) + r] = (a.L == 1) ? acc[r] : scl(acc[r], gain);
            const int64_t wb = s0 + (int64_t)R * 64 * wave;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int sl = lane + 64 * k;
                const X v = ot[sl + sl / R];
                const int64_t m = (int64_t)c + (int64_t)a.Lp * (wb + sl);
                if (m < a.n_out) y[m] = v;
            }
        } else if (a.Lp == 1 && a.L == 1 && sb + R <= a.n_out && (R * sizeof(X)) % 16 == 0 &&
                   (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
            constexpr int NV = (R * (int)sizeof(X)) / 16;
            float4 *dst = reinterpret_cast<float4 *>(y + sb);
            const float4 *src = reinterpret_cast<const float4 *>(acc);
#pragma unroll
            for (int v4 = 0; v4 < NV; ++v4) dst[v4] = src[v4];
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t m = (int64_t)c + (int64_t)a.Lp * (sb + r);
                if (m < a.n_out) y[m] = (a.L == 1) ? acc[r] : scl(acc[r], gain);
            }
        }
    }
    if (a.out_tile) {
        // Lp > 4 (e.g. the reference's default L = 12): a per-class store would fill 8 of every 8*Lp
        // bytes per pass (0.72 TB/s at L = 12); the staged image goes out in 16-byte pieces, 1 KiB per
        // wave-instruction
        __syncthreads();
        const int64_t m0 = (int64_t)a.Lp * s0;
        const int64_t total = (int64_t)a.Lp * 256 * R;
        constexpr int VEC = 16 / (int)sizeof(X) > 0 ? 16 / (int)sizeof(X) : 1;
        const char *tile = smem_raw + a.out_off;
        if (m0 + total <= a.n_out && (total % VEC) == 0 && ((reinterpret_cast<uintptr_t>(y + m0)) & 15) == 0) {
            const float4 *src = reinterpret_cast<const float4 *>(tile);
            float4 *dst = reinterpret_cast<float4 *>(y + m0);
            for (int i = tid; i < (int)(total / VEC); i += 256) dst[i] = src[i];
        } else {
            const X *src = reinterpret_cast<const X *>(tile);
            for (int64_t i = tid; i < total; i += 256)
                if (m0 + i < a.n_out) y[m0 + i] = src[i];
        }
    }
    recheck_outputs(a.Lp);
}

// ------------------------------------------------------------------ host side
FirHandle::~FirHandle()
{
    for (auto &p : poly) if (p.dev) (void)hipFree(p.dev);
    for (auto &t : sw) { if (t.taps) (void)hipFree(t.taps); if (t.rho) (void)hipFree(t.rho); }
    for (auto &t : mm) if (t.At) (void)hipFree(t.At);
    for (auto &t : bx) if (t.At) (void)hipFree(t.At);
    if (ols) fir_ols_free(ols);
    for (auto &u : ols_up) fir_ols_free(u.plan);
    if (up4k) fir_up4k_free(up4k);
    if (up2k) fir_up2k_free(up2k);
    if (dn4k) fir_dn4k_free(dn4k);
    if (ols64) fir_ols64_free(ols64);
    for (auto &u : ols64_up) fir_ols64_free(u.plan);
    for (FirHandle *p : parts) delete p;
    for (FirHandle *p : heads) delete p;
    if (taps64_dev) (void)hipFree(taps64_dev);
}

int fir_careful(FirHandle *h, CarefulFir *out)
{
    if (!h->taps64_dev) {
        const size_t bytes = h->taps_host.size() * sizeof(double);
        SK_HIP(hipMalloc(&h->taps64_dev, bytes));
        SK_HIP(hipMemcpy(h->taps64_dev, h->taps_host.data(), bytes, hipMemcpyHostToDevice));
    }
    out->taps = (const double *)h->taps64_dev;
    out->ntaps = h->ntaps;
    out->taps_complex = h->taps_complex ? 1 : 0;
    return SKDSP_OK;
}

// polyphase bank for interpolation factor L in the compute precision
static int get_bank(FirHandle *h, int L, void **dev, int *T_out)
{
    for (auto &p : h->poly)
        if (p.L == L) { *dev = p.dev; *T_out = p.T; return SKDSP_OK; }
    const int P = h->ntaps;
    const int T = (P + L - 1) / L;
    const bool dbl = dtype_double(h->dtype);
    const int comp = h->taps_complex ? 2 : 1;
    const size_t esz = (dbl ? 8 : 4) * comp;
    std::vector<char> host((size_t)L * T * esz, 0);
    for (int phi = 0; phi < L; ++phi)
        for (int t = 0; t < T; ++t) {
            const int k = phi + L * t;
            if (k >= P) continue;
            for (int cc = 0; cc < comp; ++cc) {
                const double v = h->taps_host[(size_t)k * comp + cc];
                const size_t idx = ((size_t)phi * T + t) * comp + cc;
                if (dbl) reinterpret_cast<double *>(host.data())[idx] = v;
                else reinterpret_cast<float *>(host.data())[idx] = (float)v;
            }
        }
    void *d = nullptr;
    SK_HIP(hipMalloc(&d, host.size()));
    SK_HIP(hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
    h->poly.push_back({L, T, d});
    *dev = d;
    *T_out = T;
    return SKDSP_OK;
}

template <typename X, typename B>
static int launch_typed(const X *x, const B *bank, const PolyArgs &a, int R, int nblocks, size_t lds, X *y, hipStream_t s)
{
    switch (R) {
    case 1: hipLaunchKernelGGL((fir_poly_kernel<X, B, 1>), dim3(nblocks), dim3(256), lds, s, x, bank, a, y); break;
    case 2: hipLaunchKernelGGL((fir_poly_kernel<X, B, 2>), dim3(nblocks), dim3(256), lds, s, x, bank, a, y); break;
    default: hipLaunchKernelGGL((fir_poly_kernel<X, B, 4>), dim3(nblocks), dim3(256), lds, s, x, bank, a, y); break;
    }
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

// zero-padded per-(class, residue) tap tables for the sliding-window kernel:
//   taps[(c q + j) nB R + m'] = b[phi_c + L (q (m' - delta) + j)]   (0 outside),  rho[c q + j] in [0,q) or -1
static int get_sw_table(FirHandle *h, int L, int M, int R, int nB, const FirHandle::SwTab **out)
{
    for (auto &t : h->sw)
        if (t.L == L && t.M == M && t.R == R) { *out = &t; return SKDSP_OK; }
    const int g = std::gcd(L, M), Lp = L / g, q = M / g;
    const int P = h->ntaps, T = (P + L - 1) / L;
    const bool dbl = dtype_double(h->dtype);
    const int comp = h->taps_complex ? 2 : 1;
    const size_t esz = (dbl ? 8 : 4) * comp;
    const size_t pitch = (size_t)nB * R;
    std::vector<char> host((size_t)Lp * q * pitch * esz, 0);
    std::vector<int> rho((size_t)Lp * q, -1);
    for (int c = 0; c < Lp; ++c) {
        const long long cm = (long long)c * M;
        const int phi = (int)(cm % L), ic = (int)(cm / L);
        for (int j = 0; j < q; ++j) {
            if (j >= T) continue;
            const int delta = ic < j ? 1 : 0;
            rho[(size_t)c * q + j] = ic - j + q * delta;
            const int Tj = (T - j + q - 1) / q;
            for (int m = 0; m < Tj; ++m) {
                const int t = q * m + j;          // tap index inside the phase
                const int k = phi + L * t;        // original tap
                if (k >= P) continue;
                const size_t idx0 = ((size_t)c * q + j) * pitch + (size_t)(m + delta);
                for (int cc = 0; cc < comp; ++cc) {
                    const double v = h->taps_host[(size_t)k * comp + cc];
                    if (dbl) reinterpret_cast<double *>(host.data())[idx0 * comp + cc] = v;
                    else reinterpret_cast<float *>(host.data())[idx0 * comp + cc] = (float)v;
                }
            }
        }
    }
    FirHandle::SwTab t;
    t.L = L; t.M = M; t.R = R; t.taps = nullptr; t.rho = nullptr;
    SK_HIP(hipMalloc(&t.taps, host.size()));
    SK_HIP(hipMalloc(&t.rho, rho.size() * sizeof(int)));
    SK_HIP(hipMemcpy(t.taps, host.data(), host.size(), hipMemcpyHostToDevice));
    SK_HIP(hipMemcpy(t.rho, rho.data(), rho.size() * sizeof(int), hipMemcpyHostToDevice));
    h->sw.push_back(t);
    *out = &h->sw.back();
    return SKDSP_OK;
}

int fir_direct_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, int L, int M, int64_t n_out, void *y,
                      hipStream_t s)
{
    SK_CHECK(L >= 1 && M >= 1, SKDSP_ERR_BADARG, "fir: L and M must be >= 1 (L=%d M=%d)", L, M);
    if (n_out <= 0) return SKDSP_OK;
    SK_CHECK(!(h->taps_complex && !dtype_complex(h->dtype)), SKDSP_ERR_BADARG,
             "fir: complex taps need a complex signal dtype (promote x first)");
    const int mm_mode = opt().fir_mm, bx_mode = opt().fir_bx;  // 0: never (developer A/B)
    if (mm_mode && bx_mode && fir_bx_supported(h, L, M, n_out)) return fir_bx_launch(h, x, n, n_hist, L, M, n_out, y, s);
    if (mm_mode && fir_mm_supported(h, L, M, n_out)) return fir_mm_launch(h, x, n, n_hist, L, M, n_out, y, s);
    void *bank = nullptr;
    int T = 0;
    int rc = get_bank(h, L, &bank, &T);
    if (rc) return rc;
    note_path("fir_direct");

    const int g = std::gcd(L, M);
    PolyArgs a;
    a.n = n; a.n_hist = n_hist; a.n_out = n_out;
    a.T = T; a.L = L; a.M = M; a.Lp = L / g; a.q = M / g;
    a.n_s = (n_out + a.Lp - 1) / a.Lp;
    if ((rc = fir_careful(h, &a.cf))) return rc;

    const size_t esz = dtype_size(h->dtype);
    const size_t lds_cap = 80 * 1024;  // <= 2 workgroups per CU of the 160 KiB LDS

    // ---- preferred: register sliding window (R consecutive outputs per thread) ----
    const bool no_sw = false;
    if (!no_sw) {
        const int Rmax = dtype_double(h->dtype) && dtype_complex(h->dtype) ? 4 : 8;
        for (int R = Rmax; R >= 2; R >>= 1) {
            const int q = a.q, P = q * R;
            const int Tq = (T + q - 1) / q;
            const int nB = (Tq + 1 + R - 1) / R;      // +1: the optional leading zero tap (delta)
            const int G = nB;
            const int64_t win = (int64_t)(G + 256) * P;
            const int64_t phys = (int64_t)(G + 256) * (P + 1);
            if ((size_t)phys * esz > lds_cap) continue;
            const int64_t nb = (a.n_s + 256 * R - 1) / (256 * R);
            if (nb < ctx().num_cus && R > 2) continue;  // small problems: smaller tiles
            // interpolation by more than 4 classes: the whole output run of the workgroup is staged
            // in LDS (needs 256*R*Lp elements next to the window) -- prefer a smaller R that fits
            const size_t tile_bytes = (size_t)256 * R * a.Lp * esz;
            const bool want_tile = a.Lp > 4;  // up to 4 classes the per-class row transposition measured faster
            const bool tile_fits = (size_t)phys * esz + 4096 + tile_bytes <= lds_cap;
            if (want_tile && !tile_fits && R > 2) continue;
            const FirHandle::SwTab *tab = nullptr;
            int rc2 = get_sw_table(h, L, M, R, nB, &tab);
            if (rc2) return rc2;
            SwArgs w;
            w.n = n; w.n_hist = n_hist; w.n_out = n_out;
            w.L = L; w.Lp = a.Lp; w.q = q; w.G = G; w.nB = nB; w.win = (int)win;
            w.M = M; w.cf = a.cf;
            w.half_last = (R == 8 && (Tq + 1) - (nB - 1) * R <= R / 2) ? 1 : 0;
            size_t lds = (size_t)phys * esz;
            w.tap_off = -1; w.tap_cnt = 0;
            if (h->dtype == SKDSP_C64 && !h->taps_complex && R == 8) {
                const size_t tbytes = (size_t)q * nB * R * 4;  // one class; two buffers
                const size_t off = (lds + 15) & ~(size_t)15;
                if (off + 2 * tbytes <= lds_cap) { w.tap_off = (int)off; w.tap_cnt = (int)(tbytes / 4); lds = off + 2 * tbytes; }
            }
            w.out_off = -1;
            w.out_tile = 0;
            const bool lpt_ok = h->dtype == SKDSP_C64 && !h->taps_complex && R == 8 && w.tap_off >= 0 && a.Lp >= 2 && a.Lp <= 4;
            if (lpt_ok) {
                w.out_off = 0;  // unrolled-class kernel: its output tiles alias the (finished) window image
            } else if (want_tile && tile_fits) {
                const size_t off = (lds + 15) & ~(size_t)15;
                w.out_off = (int)off;
                w.out_tile = 1;
                lds = off + tile_bytes;
            } else {
                const size_t off = (lds + 15) & ~(size_t)15;
                const size_t obytes = (size_t)4 * 64 * (R + 1) * esz;
                if (a.Lp > 1 && off + obytes <= lds_cap) { w.out_off = (int)off; lds = off + obytes; }  // Lp == 1 stores 16-byte vectors directly
            }
#define SK_SWQ(XT, BT, RR, QQ)                                                                                        \
    do {                                                                                                              \
        if (lds > 64 * 1024)                                                                                          \
            (void)hipFuncSetAttribute((const void *)fir_sw_kernel<XT, BT, RR, QQ>,                                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        hipLaunchKernelGGL((fir_sw_kernel<XT, BT, RR, QQ>), dim3((unsigned)nb), dim3(256), lds, s, (const XT *)x,     \
                           (const BT *)tab->taps, (const int *)tab->rho, w, (XT *)y);                                 \
    } while (0)
#define SK_SWR(XT, BT, RR)                                   \
    do {                                                     \
        if (q == 1) SK_SWQ(XT, BT, RR, 1);                   \
        else if (q == 2) SK_SWQ(XT, BT, RR, 2);              \
        else if (q == 3) SK_SWQ(XT, BT, RR, 3);              \
        else SK_SWQ(XT, BT, RR, 0);                          \
    } while (0)
#define SK_SW(XT, BT)                                        \
    do {                                                     \
        if (R == 8) SK_SWR(XT, BT, 8);                       \
        else if (R == 4) SK_SWR(XT, BT, 4);                  \
        else SK_SWR(XT, BT, 2);                              \
    } while (0)
#define SK_SWL(QQ, LL)                                                                                                 \
    do {                                                                                                               \
        if (lds > 64 * 1024)                                                                                           \
            (void)hipFuncSetAttribute((const void *)fir_sw_kernel<float2, float, 8, QQ, LL>,                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
        hipLaunchKernelGGL((fir_sw_kernel<float2, float, 8, QQ, LL>), dim3((unsigned)nb), dim3(256), lds, s,           \
                           (const float2 *)x, (const float *)tab->taps, (const int *)tab->rho, w, (float2 *)y);        \
    } while (0)
#define SK_SWLQ(LL)                                   \
    do {                                              \
        if (q == 1) SK_SWL(1, LL);                    \
        else if (q == 2) SK_SWL(2, LL);               \
        else if (q == 3) SK_SWL(3, LL);               \
        else SK_SWL(0, LL);                           \
    } while (0)
            if (lpt_ok) {
                if (a.Lp == 2) SK_SWLQ(2);
                else if (a.Lp == 3) SK_SWLQ(3);
                else SK_SWLQ(4);
                SK_HIP(hipGetLastError());
                return SKDSP_OK;
            }
#undef SK_SWLQ
#undef SK_SWL
            switch (h->dtype) {
            case SKDSP_F32: SK_SW(float, float); break;
            case SKDSP_F64: SK_SW(double, double); break;
            case SKDSP_C64: if (h->taps_complex) SK_SW(float2, float2); else SK_SW(float2, float); break;
            case SKDSP_C128: if (h->taps_complex) SK_SW(double2, double2); else SK_SW(double2, double); break;
            }
#undef SK_SW
#undef SK_SWR
#undef SK_SWQ
            SK_HIP(hipGetLastError());
            return SKDSP_OK;
        }
    }

    // ---- generic fallback: one LDS read per FMA, any stride ----
    const int64_t cap_elems = (int64_t)((size_t)64 * 1024 / esz);
    // window = q*s_tile + T + q  <=  cap_elems
    int64_t s_tile = (cap_elems - T - a.q) / a.q;
    SK_CHECK(s_tile >= 1, SKDSP_ERR_UNSUPPORTED,
             "fir_direct: %d taps/phase with stride %d do not fit the 64 KiB LDS window", T, a.q);
    int R = 4;
    if (s_tile > 1024) s_tile = 1024;
    if (a.n_s < s_tile) s_tile = a.n_s;
    if (s_tile <= 256) R = 1; else if (s_tile <= 512) R = 2;
    // spread small problems over more workgroups
    while (R > 1 && (a.n_s + s_tile - 1) / s_tile < 2 * ctx().num_cus) { R >>= 1; s_tile = (s_tile > 256 * R) ? 256 * R : s_tile; }
    a.s_tile = (int)s_tile;
    a.win = (int)(a.q * s_tile + T + a.q);
    const size_t lds = (size_t)a.win * esz;
    const int nblocks = (int)((a.n_s + s_tile - 1) / s_tile);

    switch (h->dtype) {
    case SKDSP_F32: return launch_typed((const float *)x, (const float *)bank, a, R, nblocks, lds, (float *)y, s);
    case SKDSP_F64: return launch_typed((const double *)x, (const double *)bank, a, R, nblocks, lds, (double *)y, s);
    case SKDSP_C64:
        if (h->taps_complex) return launch_typed((const float2 *)x, (const float2 *)bank, a, R, nblocks, lds, (float2 *)y, s);
        return launch_typed((const float2 *)x, (const float *)bank, a, R, nblocks, lds, (float2 *)y, s);
    case SKDSP_C128:
        if (h->taps_complex) return launch_typed((const double2 *)x, (const double2 *)bank, a, R, nblocks, lds, (double2 *)y, s);
        return launch_typed((const double2 *)x, (const double *)bank, a, R, nblocks, lds, (double2 *)y, s);
    }
    SK_CHECK(false, SKDSP_ERR_BADARG, "fir: bad dtype %d", h->dtype);
}

}  // namespace skdsp
