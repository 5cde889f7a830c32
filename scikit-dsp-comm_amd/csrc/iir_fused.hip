// iir_fused.hip -- single-pass exact IIR scan for gfx950 (MI355X): one launch, the signal is read once and written once.
// Serves scipy.signal.sosfilt(sos, x) (multirate_helper.py:173) / lfilter(b, a, x) (:74, :81) for real signals and
// decaying cascades of up to 8 biquads; everything else takes the K1 / carries / K3 path of iir_scan.hip.
#include "iir_common.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>


namespace skdsp {

// ---- the chunk-level scan on the FP64 matrix pipe ---------------------------------------------------------------------
// Phase A leaves the from-rest end states of a wave's 64 chunks as four accumulator tiles of v_mfma_f64_16x16x4: tile g,
// register r, lane t holds state row (t >> 4) + 4 r of chunk column 16 g + (t & 15).  Register r of such a tile is, lane
// for lane, the B operand of reduction step r of the same instruction (B[k = 4 r + (t >> 4)][col = t & 15]), so one
// Hillis-Steele level  v_j += M^(2^l) v_(j - 2^l)  is four MFMAs per tile with the level's matrix as A operand and the
// column-shifted OLD tiles as B: shifts below 16 columns are DPP row rotations (+ a select for the columns that come from
// the tile on the left), shifts by 16 / 32 columns just name another tile.  No LDS, no barrier and no scalar matrix loads
// for the six levels inside a wave; the VALU version (matvec_acc per thread, 144 FMAs per level, matrices through the
// scalar cache, two barriers per level) took 9.7 + 5.8 us of a 53 us segment (tools/fused_trace.py) and a quarter of the
// kernel's FP64 VALU instructions.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)  // lanes without a source read 0
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
constexpr int kDppRowShr = 0x110, kDppRowRor = 0x120;


// tiles[g] += M_L * (tiles shifted right by 2^L columns inside the wave), L < 4;  A = the level's matrix as A operands
// (IirPlan::pwa_dev, [level][step][64]: read from global memory one level ahead of their use)
struct AOps { double a[4]; };
__device__ __forceinline__ AOps load_aops(const double *__restrict__ pwa, int level, int lane)
{
    AOps o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o.a[r] = pwa[(level * 4 + r) * 64 + lane];  // 512 contiguous bytes per wave and step, L1 / L2 resident
    return o;
}
template <int L>
__device__ __forceinline__ void scan_level_rows(v4d_t (&t)[4], const AOps &A, int lane)
{
    constexpr int S = 1 << L;
    const bool own = (lane & 15) >= S;
    double rot[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) rot[g][r] = dpp_f64<kDppRowRor + S>(t[g][r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double a = A.a[r];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const double b = own ? rot[g][r] : (g > 0 ? rot[g - 1][r] : 0.0);
            t[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, t[g], 0, 0, 0);
        }
    }
}
// one tile: t += M_L * (t shifted right by SH columns, zero fill)
template <int SH>
__device__ __forceinline__ void tile_level(v4d_t &t, const AOps &A)
{
    double sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[r] = dpp_f64<kDppRowShr + SH>(t[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) t = __builtin_amdgcn_mfma_f64_16x16x4f64(A.a[r], sh[r], t, 0, 0, 0);
}
// dst += M_level * src (whole tiles)
__device__ __forceinline__ void tile_mul_acc(v4d_t &dst, const v4d_t &src, const AOps &A)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) dst = __builtin_amdgcn_mfma_f64_16x16x4f64(A.a[r], src[r], dst, 0, 0, 0);
}

// ------------------------------------------------------------------ single-pass scan
// One launch, x read once, y written once (decaying cascades of <= 8 biquads, real signals).
//
// A workgroup owns one SEGMENT of 256 chunks x T samples (T = 128 float32 / 64 float64 samples: 32768 / 16384 samples)
// and keeps every sample of it ON CHIP between the two sweeps an exact scan needs -- each thread holds its own chunk in
// registers (128 VGPRs):
//   A  the segment streams in through a [256 rows x 32 samples] LDS image (full-line 16-byte loads, next piece in
//      flight); every thread copies its row into registers, and the same image feeds the FP64 matrix pipe with the B
//      operands of  V = G x  (the from-rest end state of each chunk, as in iir_k1r_kernel; G from LDS)
//   S  from-rest Hillis-Steele scan of the 256 chunk states -> p_j; p_255 is the segment's end state from rest
//   L  decoupled look-back: the segment publishes p_255 as 8-byte {half, epoch} granules (relaxed agent-scope atomics:
//      the data is its own flag, no fence) and reads its predecessor's.  For a filter whose transition over one
//      segment is below 1e-30 (the applicability test) the predecessor's from-rest end state IS the exact state c at
//      the start of this segment, so no workgroup ever waits for a chain.  Segments are handed out by a ticket
//      counter, so a predecessor is always a workgroup that already runs; the poll is bounded and reports through
//      a.err instead of hanging.
//   C  exact initial state of chunk j:  z_j = p_(j-1) + M^j c, M^j c by binary powers (only chunks below 2^n_lv)
//   B  the recurrence over the 128 register-resident samples, outputs written back through the LDS image as full lines
// Per sample: 8 B of HBM traffic (the algorithmic bytes), ~40 FP64 VALU instructions of recurrence + ~12 of scan.
struct FusedArgs {
    const void *x;
    void *y;
    int64_t n;
    int64_t batch_stride;
    const double *pw;            // M^(2^l), M = A^T, row-major D x D each
    const double *gt;            // G in MFMA A-operand order [T/4][64]
    unsigned long long *lb;      // [batch][nseg][32] look-back granules
    unsigned long long *ticket;  // [batch] segment dispenser (monotonic; ticket_base = its value before this launch)
    unsigned long long ticket_base[2];
    unsigned epoch;
    int nseg;
    int n_lv;
    const double *zi;            // [batch][D] or null
    double *zf;                  // [batch][D] or null
    unsigned *err;               // set to 1 if a look-back poll gave up
    int dec;                     // > 1: only y[k * dec] is stored (at y[k]), k < n_keep / dec  (.dn: no full-rate result in HBM)
    int dec_dq, dec_dr;          // (rows between a thread's staged segments x T) div / mod dec
    int64_t n_keep;              // (n / dec) * dec
};

template <int NSEC, typename IO, bool UNIT>
__global__ __launch_bounds__(kIirThreads, 2) void iir_fused_kernel(FusedArgs a, Coef<NSEC, 2> cf, const double *__restrict__ pw,
                                                                   const double *__restrict__ gtab, const double *__restrict__ pwa)
{
    // (pw / gtab are separate __restrict__ parameters: the ticket atomic and the look-back stores precede the scan, and
    // only loads through a noalias pointer stay scalar loads behind them -- as plain members of `a` every matrix
    // entry of every scan level became a per-lane global load: 6.3 M vector loads per launch instead of 0.33 M)
    constexpr int ORD = 2, D = NSEC * ORD;
    constexpr bool MSCAN = NSEC >= 6;
    constexpr int T = 512 / (int)sizeof(IO);
    constexpr int NP = T / kPiece;
    using St = Stage<IO>;
    constexpr int kStageBytes = kIirThreads * St::pitch * (int)sizeof(IO);
    constexpr int kScanBytes = kIirThreads * D * 8;
    constexpr int kLdsBytes = kStageBytes > kScanBytes ? kStageBytes : kScanBytes;
    __shared__ __attribute__((aligned(16))) char lds_raw[kLdsBytes];
    __shared__ double gl[(T / 4) * 64];
    __shared__ double qsh[4 * 16], esh[4 * 16], aggsh[16];
    __shared__ unsigned cw[32];
    __shared__ int seg_sh;
    IO *stage = reinterpret_cast<IO *>(lds_raw);
    double *sc = reinterpret_cast<double *>(lds_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bat = blockIdx.y;
    if (tid == 0) seg_sh = (int)(atomicAdd(a.ticket + bat, 1ull) - a.ticket_base[bat]);
    for (int i = tid; i < (T / 4) * 64; i += kIirThreads) gl[i] = gtab[i];
    __syncthreads();
    const int seg = seg_sh;
    const IO *x = reinterpret_cast<const IO *>(a.x) + (size_t)bat * a.batch_stride;
    IO *y = reinterpret_cast<IO *>(a.y) + (size_t)bat * a.batch_stride;
    const int64_t row0 = (int64_t)seg * kIirThreads;   // first chunk of the segment
    const bool interior = (row0 + kIirThreads) * T <= a.n;

    typedef float pre_t __attribute__((ext_vector_type(4)));
    pre_t pre[St::per_thread];
    auto load_piece = [&](int p) {  // interior segments only
#pragma unroll
        for (int i = 0; i < St::per_thread; ++i) {
            const int idx = i * 64 + lane;
            const int row = wave * 64 + idx / St::segs, sg = idx % St::segs;
            const int64_t g = (row0 + row) * T + (int64_t)p * kPiece + (int64_t)sg * St::elems;
            pre[i] = __builtin_nontemporal_load(reinterpret_cast<const pre_t *>(x + g));
        }
    };
    auto stage_slow = [&](int p) {  // zero beyond the signal
#pragma unroll 1
        for (int i = 0; i < St::per_thread; ++i) {
            const int idx = i * 64 + lane;
            const int row = wave * 64 + idx / St::segs, sg = idx % St::segs;
            const int64_t g = (row0 + row) * T + (int64_t)p * kPiece + (int64_t)sg * St::elems;
            IO *dst = stage + row * St::pitch + sg * St::elems;
#pragma unroll
            for (int e = 0; e < St::elems; ++e) dst[e] = (g + e < a.n) ? x[g + e] : IO(0);
        }
    };

    // ---- A: stream the segment in; chunk rows to registers; V = G x on the matrix pipe --------------------------
    // (every wave stages, transposes and multiplies ITS OWN 64 rows: phases A and B need no workgroup barrier, the four waves
    // drift apart and one's LDS work runs under another's MFMAs or recurrence)
    IO xr[T];
    v4d_t acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = v4d_t{0.0, 0.0, 0.0, 0.0};
    const int c = lane & 15, j = lane >> 4;
    IO *myrow = stage + tid * St::pitch;
    if (interior) load_piece(0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (interior) {
#pragma unroll
            for (int i = 0; i < St::per_thread; ++i) {
                const int idx = i * 64 + lane;
                const int row = wave * 64 + idx / St::segs, sg = idx % St::segs;
                *reinterpret_cast<pre_t *>(stage + row * St::pitch + sg * St::elems) = pre[i];
            }
        } else {
            stage_slow(p);
        }
        wave_lds_sync();
        if (interior && p + 1 < NP) load_piece(p + 1);  // in flight during the MFMAs
#pragma unroll
        for (int sgi = 0; sgi < St::segs; ++sgi) {
            const float4 raw = *reinterpret_cast<const float4 *>(myrow + sgi * St::elems);
            const IO *e4 = reinterpret_cast<const IO *>(&raw);
#pragma unroll
            for (int e = 0; e < St::elems; ++e) xr[p * kPiece + sgi * St::elems + e] = e4[e];
        }
        const IO *xs = stage + (wave * 64 + c) * St::pitch + j;
#pragma unroll
        for (int s = 0; s < kPiece / 4; ++s) {
            const double ga = gl[(p * (kPiece / 4) + s) * 64 + lane];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga, (double)xs[g * 16 * St::pitch + 4 * s], acc[g], 0, 0, 0);
        }
        wave_lds_sync();
    }
    __syncthreads();  // the scan's exchange array lies over every wave's rows

    double z[D];
    // The scan of the 256 chunk maps: on the matrix pipe for cascades of 6+ biquads (a 12 x 12 .. 16 x 16 transition: 84+
    // FMAs per thread and level on the VALU), per thread on the VALU for smaller ones (the MFMA tiles are 16 x 16 whatever D
    // is: one biquad 0.121 -> 0.147 ms, order 8 0.138 -> 0.148 ms with them; 8 biquads 0.197 -> 0.182 ms, float64 0.321 -> 0.277)
    if constexpr (MSCAN) {
        // ---- S: from-rest inclusive scan of each wave's 64 chunk maps, on the matrix pipe (scan_level_rows) ----------------
        {
            AOps A0 = load_aops(pwa, 0, lane), A1 = load_aops(pwa, 1, lane);
            if (a.n_lv > 0) scan_level_rows<0>(acc, A0, lane);
            A0 = load_aops(pwa, 2, lane);
            if (a.n_lv > 1) scan_level_rows<1>(acc, A1, lane);
            A1 = load_aops(pwa, 3, lane);
            if (a.n_lv > 2) scan_level_rows<2>(acc, A0, lane);
            A0 = load_aops(pwa, 4, lane);
            if (a.n_lv > 3) scan_level_rows<3>(acc, A1, lane);
            A1 = load_aops(pwa, 5, lane);
            if (a.n_lv > 4) {
    #pragma unroll
                for (int g = 3; g >= 1; --g) tile_mul_acc(acc[g], acc[g - 1], A0);   // descending: the right-hand tiles are still old
            }
            if (a.n_lv > 5) {
                tile_mul_acc(acc[3], acc[1], A1);
                tile_mul_acc(acc[2], acc[0], A1);
            }
        }
        AOps A6 = {}, A7 = {};
    if (a.n_lv > 6) {   // (a filter that remembers more than 64 chunks: most do not)
        A6 = load_aops(pwa, 6, lane);
        A7 = load_aops(pwa, 7, lane);
    }
        // across the four waves: Q_w = the wave's last column; P = inclusive scan of (Q_0 .. Q_3) with M^64 (levels 6, 7) as a
        // 4-column tile, every wave for itself
        if (c == 15) {
    #pragma unroll
            for (int r = 0; r < 4; ++r) qsh[wave * 16 + j + 4 * r] = acc[3][r];
        }
        __syncthreads();
        v4d_t pm;
    #pragma unroll
        for (int r = 0; r < 4; ++r) pm[r] = c < 4 ? qsh[c * 16 + j + 4 * r] : 0.0;
        if (a.n_lv > 6) tile_level<1>(pm, A6);
        if (a.n_lv > 7) tile_level<2>(pm, A7);
        if (wave == 0 && c == 3) {
    #pragma unroll
            for (int r = 0; r < 4; ++r) aggsh[j + 4 * r] = pm[r];   // the segment's end state from rest
        }
        __syncthreads();

        // ---- L: publish the segment's end state from rest, fetch the predecessor's ------------------------------------
        if (tid < 2 * D) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(aggsh[tid >> 1]);
            const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
            unsigned long long *slot = a.lb + ((size_t)bat * a.nseg + seg) * 32 + tid;
            __hip_atomic_store(slot, ((unsigned long long)a.epoch << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned got = 0;
            if (seg > 0) {
                const unsigned long long *src = a.lb + ((size_t)bat * a.nseg + seg - 1) * 32 + tid;
                unsigned long long g = 0;
                int spins = 0;
                for (;;) {
                    g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(g >> 32) == a.epoch) break;
                    if (++spins > (1 << 22)) {  // ~ seconds: never in a healthy run; fail loudly instead of hanging the GPU
                        *a.err = 1u;
                        g = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                got = (unsigned)g;
            } else if (a.zi) {
                const unsigned long long zb = (unsigned long long)__double_as_longlong(a.zi[(size_t)bat * D + (tid >> 1)]);
                got = (tid & 1) ? (unsigned)(zb >> 32) : (unsigned)zb;
            }
            cw[tid] = got;
        } else if (tid < 32) {
            cw[tid] = 0u;
        }
        __syncthreads();

        // ---- C: the exact state e_w at the start of wave w = M^(64 w) c + P_(w-1); correction columns M^(jl+1) e_w by doubling;
        //         z_j = (scan + correction) of the column to the left -----------------------------------------------------------
        v4d_t cm;
    #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = j + 4 * r;
            cm[r] = c == 0 ? __longlong_as_double((long long)(((unsigned long long)cw[2 * d + 1] << 32) | cw[2 * d])) : 0.0;
        }
        if (a.n_lv > 6) tile_level<1>(cm, A6);   // column w: M^(64 w) c
        if (a.n_lv > 7) tile_level<2>(cm, A7);
        if (c == wave) {
    #pragma unroll
            for (int r = 0; r < 4; ++r) esh[wave * 16 + j + 4 * r] = cm[r];
        }
        {
            const double p_left0 = dpp_f64<kDppRowShr + 1>(pm[0]), p_left1 = dpp_f64<kDppRowShr + 1>(pm[1]);
            const double p_left2 = dpp_f64<kDppRowShr + 1>(pm[2]), p_left3 = dpp_f64<kDppRowShr + 1>(pm[3]);
            if (c == wave) {
                esh[wave * 16 + j] += p_left0;
                esh[wave * 16 + j + 4] += p_left1;
                esh[wave * 16 + j + 8] += p_left2;
                esh[wave * 16 + j + 12] += p_left3;
            }
        }
        __builtin_amdgcn_wave_barrier();  // (esh[wave] is written and read by this wave only: LDS serves a wave in order)
        v4d_t u[4];
        {
            const v4d_t zero = {0.0, 0.0, 0.0, 0.0};
            v4d_t e0;
    #pragma unroll
            for (int r = 0; r < 4; ++r) e0[r] = c == 0 ? esh[wave * 16 + j + 4 * r] : 0.0;
            u[0] = u[1] = u[2] = u[3] = zero;
            AOps A0 = load_aops(pwa, 0, lane), A1 = load_aops(pwa, 1, lane);
            tile_mul_acc(u[0], e0, A0);                    // column 0: M e_w
            if (a.n_lv > 0) tile_level<1>(u[0], A0);       // columns [2^l, 2^(l+1)) = M^(2^l) * columns [0, 2^l)
            A0 = load_aops(pwa, 2, lane);
            if (a.n_lv > 1) tile_level<2>(u[0], A1);
            A1 = load_aops(pwa, 3, lane);
            if (a.n_lv > 2) tile_level<4>(u[0], A0);
            A0 = load_aops(pwa, 4, lane);
            if (a.n_lv > 3) tile_level<8>(u[0], A1);
            A1 = load_aops(pwa, 5, lane);
            if (a.n_lv > 4) tile_mul_acc(u[1], u[0], A0);
            if (a.n_lv > 5) {
                tile_mul_acc(u[2], u[0], A1);
                tile_mul_acc(u[3], u[1], A1);
            }
        }
    #pragma unroll
        for (int g = 0; g < 4; ++g)
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = j + 4 * r;
                if (d < D) sc[d * kIirThreads + wave * 64 + g * 16 + c] = acc[g][r] + u[g][r];
            }
        __syncthreads();
    #pragma unroll
        for (int d = 0; d < D; ++d) z[d] = lane ? sc[d * kIirThreads + tid - 1] : esh[wave * 16 + d];

    } else {
        // chunk end states from the accumulator layout (col = lane & 15, row = (lane >> 4) + 4 reg) to one thread per chunk
    #pragma unroll
        for (int g = 0; g < 4; ++g)
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = j + 4 * r;
                if (d < D) sc[d * kIirThreads + wave * 64 + g * 16 + c] = acc[g][r];
            }
        __syncthreads();
        double v[D];
    #pragma unroll
        for (int d = 0; d < D; ++d) v[d] = sc[d * kIirThreads + tid];
        __syncthreads();

        // ---- S: from-rest inclusive scan of the 256 chunk maps ---------------------------------------------------------
    #pragma unroll 1
        for (int l = 0; l < a.n_lv; ++l) {
            const int s = 1 << l;
    #pragma unroll
            for (int d = 0; d < D; ++d) sc[d * kIirThreads + tid] = v[d];
            __syncthreads();
            if (tid >= s) {
                double left[D];
    #pragma unroll
                for (int d = 0; d < D; ++d) left[d] = sc[d * kIirThreads + tid - s];
                matvec_acc<D, ORD>(pw + (size_t)l * D * D, left, v);
            }
            __syncthreads();
        }
    #pragma unroll
        for (int d = 0; d < D; ++d) sc[d * kIirThreads + tid] = v[d];
        __syncthreads();
    #pragma unroll
        for (int d = 0; d < D; ++d) z[d] = tid ? sc[d * kIirThreads + tid - 1] : 0.0;

        // ---- L: publish the segment's end state from rest, fetch the predecessor's ------------------------------------
        if (tid < 2 * D) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(sc[(tid >> 1) * kIirThreads + kIirThreads - 1]);
            const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
            unsigned long long *slot = a.lb + ((size_t)bat * a.nseg + seg) * 32 + tid;
            __hip_atomic_store(slot, ((unsigned long long)a.epoch << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned got = 0;
            if (seg > 0) {
                const unsigned long long *src = a.lb + ((size_t)bat * a.nseg + seg - 1) * 32 + tid;
                unsigned long long g = 0;
                int spins = 0;
                for (;;) {
                    g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(g >> 32) == a.epoch) break;
                    if (++spins > (1 << 22)) {  // ~ seconds: never in a healthy run; fail loudly instead of hanging the GPU
                        *a.err = 1u;
                        g = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                got = (unsigned)g;
            } else if (a.zi) {
                const unsigned long long zb = (unsigned long long)__double_as_longlong(a.zi[(size_t)bat * D + (tid >> 1)]);
                got = (tid & 1) ? (unsigned)(zb >> 32) : (unsigned)zb;
            }
            cw[tid] = got;
        }
        __syncthreads();

        // ---- C: z_j = p_(j-1) + M^j c ------------------------------------------------------------------------------------
        if ((wave << 6) < (1 << a.n_lv)) {  // (M^j c is below 1e-30 of c for j >= 2^n_lv)
            double u[D];
    #pragma unroll
            for (int d = 0; d < D; ++d) u[d] = __longlong_as_double((long long)(((unsigned long long)cw[2 * d + 1] << 32) | cw[2 * d]));
    #pragma unroll 1
            for (int l = 0; l < a.n_lv; ++l) {
                if ((tid >> l) & 1) {
                    double t2[D];
    #pragma unroll
                    for (int d = 0; d < D; ++d) t2[d] = 0.0;
                    matvec_acc<D, ORD>(pw + (size_t)l * D * D, u, t2);
    #pragma unroll
                    for (int d = 0; d < D; ++d) u[d] = t2[d];
                }
            }
            if (tid < (1 << a.n_lv)) {
    #pragma unroll
                for (int d = 0; d < D; ++d) z[d] += u[d];
            }
        }

    }
    // ---- B: the recurrence over the register-resident chunk; outputs leave through the LDS image --------------------
    const int64_t cj = row0 + tid;
    const bool zf_owner = a.zf != nullptr && cj == (a.n - 1) / T;
    const int zf_off = (int)((a.n - 1) % T);
    const int zf_piece = (a.zf != nullptr && (a.n - 1) / T >= row0 && (a.n - 1) / T < row0 + kIirThreads) ? zf_off / kPiece : -1;  // uniform
    __syncthreads();  // (the image is free again: every wave has read its scan result)
    // ONE copy of the 32-sample body (8 biquads: ~1300 FP64 instructions, 10 KiB of code; four copies would not stay in
    // the instruction cache the two CUs share): the piece at hand always sits in xr[0 .. 31], the rest moves down behind it
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
        if (p == zf_piece && zf_owner) {
            // streaming: the ONE thread of the launch whose piece holds sample n - 1 walks a copy of its state up to that sample
            // (rolled loop, samples re-read from x: this piece of x is not overwritten before its outputs are stored below).
            // A per-sample test inside the unrolled body cost 6 scalar instructions next to the 7 .. 42 vector ones.
            double zt[D];
#pragma unroll
            for (int d = 0; d < D; ++d) zt[d] = z[d];
            const IO *xp = x + cj * T + (int64_t)p * kPiece;
#pragma unroll 1
            for (int k = 0; k <= zf_off - p * kPiece; ++k) (void)cascade_step<NSEC, ORD, UNIT>(cf, zt, (double)xp[k]);
#pragma unroll
            for (int d = 0; d < D; ++d) a.zf[(size_t)bat * D + d] = zt[d];
        }
#pragma unroll
        for (int k = 0; k < kPiece; ++k) {
            const double yv = cascade_step<NSEC, ORD, UNIT>(cf, z, (double)xr[k]);
            xr[k] = (IO)yv;
        }
        wave_lds_sync();  // the wave's rows are free (its previous piece's stores have read them)
#pragma unroll
        for (int sgi = 0; sgi < St::segs; ++sgi) {
            float4 raw;
            IO *e4 = reinterpret_cast<IO *>(&raw);
#pragma unroll
            for (int e = 0; e < St::elems; ++e) e4[e] = xr[sgi * St::elems + e];
            *reinterpret_cast<float4 *>(myrow + sgi * St::elems) = raw;
        }
#pragma unroll
        for (int k = 0; k + kPiece < T; ++k) xr[k] = xr[k + kPiece];
        wave_lds_sync();
        int64_t dq_run = 0;
        int dr_run = 0;
#pragma unroll
        for (int i = 0; i < St::per_thread; ++i) {
            const int idx = i * 64 + lane;
            const int row = wave * 64 + idx / St::segs, sg = idx % St::segs;
            const int64_t g = (row0 + row) * T + (int64_t)p * kPiece + (int64_t)sg * St::elems;
            const pre_t val = *reinterpret_cast<const pre_t *>(stage + row * St::pitch + sg * St::elems);
            if (a.dec > 1) {
                // decimating store (as in iir_chunk_kernel): segment i of this thread starts a fixed number of samples
                // after segment i - 1, so its (quotient, remainder) by dec follow from the first by adding (dec_dq, dec_dr)
                if (i == 0) {
                    dq_run = g / a.dec;
                    dr_run = (int)(g - dq_run * a.dec);
                }
                const IO *tmp = reinterpret_cast<const IO *>(&val);
                if (a.dec >= St::elems) {  // at most one kept sample per 16-byte segment
                    const int e0 = dr_run == 0 ? 0 : a.dec - dr_run;
                    if (e0 < St::elems && g + e0 < a.n_keep) {
                        IO pick = tmp[0];
#pragma unroll
                        for (int e = 1; e < St::elems; ++e) pick = (e0 == e) ? tmp[e] : pick;
                        y[dq_run + (dr_run != 0)] = pick;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < St::elems; ++e) {
                        const int t = dr_run + e;  // < dec + elems
                        const int m = (t >= a.dec) + (t >= 2 * a.dec) + (t >= 3 * a.dec) + (t >= 4 * a.dec);
                        if (t == m * a.dec && g + e < a.n_keep) y[dq_run + m] = tmp[e];
                    }
                }
                dq_run += a.dec_dq;
                dr_run += a.dec_dr;
                if (dr_run >= a.dec) { dr_run -= a.dec; ++dq_run; }
            } else if (interior) {
                __builtin_nontemporal_store(val, reinterpret_cast<pre_t *>(y + g));
            } else if (g < a.n) {
                const IO *tmp = reinterpret_cast<const IO *>(&val);
#pragma unroll
                for (int e = 0; e < St::elems; ++e)
                    if (g + e < a.n) y[g + e] = tmp[e];
            }
        }
    }
}

// ---- interleaved complex signals ---------------------------------------------------------------------------------
// A complex signal through a real-coefficient cascade is two independent real recurrences on one memory stream.  Same
// structure as iir_fused_kernel with both components in one thread: a chunk is TC complex samples (64 complex64 /
// 32 complex128: 128 VGPRs as re[] / im[]), a segment 256 chunks; the interleaved 16-byte segments are split into two real
// LDS planes on the way in (as in iir_k3c_kernel) and re-joined on the way out; the matrix pipe takes the re and the im of
// the staged samples as two B operands against the same A operand; scan, look-back (64 granules) and correction run once
// per component; the two cascades of a sample run one after the other (2 waves per SIMD for <= 8 biquads).
template <int NSEC, typename IO, bool UNIT>
__global__ __launch_bounds__(kIirThreads, 2) void iir_fused_c_kernel(FusedArgs a, Coef<NSEC, 2> cf, const double *__restrict__ pw,
                                                                     const double *__restrict__ gtab)
{
    constexpr int ORD = 2, D = NSEC * ORD;
    constexpr int E = 16 / (int)sizeof(IO);        // scalars per 16 bytes (4 / 2)
    constexpr int PC = 4 * E;                      // complex samples per staged row piece (16 / 8): 128 bytes interleaved
    constexpr int TC = 256 / (int)sizeof(IO);      // complex samples per chunk (64 / 32)
    constexpr int NP = TC / PC;                    // 4 pieces
    constexpr int PITCH = 80 / (int)sizeof(IO);    // scalars per plane row (80 bytes: conflict-free b128 reads)
    constexpr int kStageBytes = 2 * kIirThreads * 80;
    constexpr int kScanBytes = kIirThreads * D * 8;
    constexpr int kLdsBytes = kStageBytes > kScanBytes ? kStageBytes : kScanBytes;
    __shared__ __attribute__((aligned(16))) char lds_raw[kLdsBytes];
    __shared__ double gl[(TC / 4) * 64];
    __shared__ unsigned cw[64];
    __shared__ int seg_sh;
    IO *st_re = reinterpret_cast<IO *>(lds_raw);
    IO *st_im = st_re + kIirThreads * PITCH;
    double *sc = reinterpret_cast<double *>(lds_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) seg_sh = (int)(atomicAdd(a.ticket, 1ull) - a.ticket_base[0]);
    for (int i = tid; i < (TC / 4) * 64; i += kIirThreads) gl[i] = gtab[i];
    __syncthreads();
    const int seg = seg_sh;
    const IO *x = reinterpret_cast<const IO *>(a.x);
    IO *y = reinterpret_cast<IO *>(a.y);
    const int64_t row0 = (int64_t)seg * kIirThreads;
    const bool interior = (row0 + kIirThreads) * TC <= a.n;

    typedef float pre_t __attribute__((ext_vector_type(4)));
    pre_t pre[8];
    auto load_piece = [&](int p) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = i * 64 + lane;
            const int row = wave * 64 + (idx >> 3), sg = idx & 7;  // 8 x 16-byte segments per 128-byte row piece
            const int64_t g = (row0 + row) * TC + (int64_t)p * PC + (int64_t)sg * (E / 2);  // complex index
            pre[i] = __builtin_nontemporal_load(reinterpret_cast<const pre_t *>(x + 2 * g));
        }
    };
    auto split_store = [&](const IO *e, int row, int sg) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < E / 2; ++k) {
            st_re[row * PITCH + sg * (E / 2) + k] = e[2 * k];
            st_im[row * PITCH + sg * (E / 2) + k] = e[2 * k + 1];
        }
    };
    auto stage_slow = [&](int p) {
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
            const int idx = i * 64 + lane;
            const int row = wave * 64 + (idx >> 3), sg = idx & 7;
            const int64_t g = (row0 + row) * TC + (int64_t)p * PC + (int64_t)sg * (E / 2);
            IO tmp[E];
#pragma unroll
            for (int e = 0; e < E; ++e) tmp[e] = (g + e / 2 < a.n) ? x[2 * g + e] : IO(0);
            split_store(tmp, row, sg);
        }
    };

    // ---- A: stream in, rows to registers, V = G x for both components ----
    IO xr[TC], xi[TC];
    v4d_t accr[4], acci[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) accr[g] = acci[g] = v4d_t{0.0, 0.0, 0.0, 0.0};
    const int c = lane & 15, j = lane >> 4;
    IO *rr = st_re + tid * PITCH, *ri = st_im + tid * PITCH;
    if (interior) load_piece(0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (interior) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i * 64 + lane;   // every wave stages its own 64 rows (as in iir_fused_kernel)
                split_store(reinterpret_cast<const IO *>(&pre[i]), wave * 64 + (idx >> 3), idx & 7);
            }
        } else {
            stage_slow(p);
        }
        wave_lds_sync();
        if (interior && p + 1 < NP) load_piece(p + 1);
#pragma unroll
        for (int c4 = 0; c4 < PC / E; ++c4) {
            const float4 qa = *reinterpret_cast<const float4 *>(rr + c4 * E), qb = *reinterpret_cast<const float4 *>(ri + c4 * E);
            const IO *ea = reinterpret_cast<const IO *>(&qa), *eb = reinterpret_cast<const IO *>(&qb);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                xr[p * PC + c4 * E + e] = ea[e];
                xi[p * PC + c4 * E + e] = eb[e];
            }
        }
        const IO *sr = st_re + (wave * 64 + c) * PITCH + j, *si = st_im + (wave * 64 + c) * PITCH + j;
#pragma unroll
        for (int s4 = 0; s4 < PC / 4; ++s4) {
            const double ga = gl[(p * (PC / 4) + s4) * 64 + lane];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                accr[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga, (double)sr[g * 16 * PITCH + 4 * s4], accr[g], 0, 0, 0);
                acci[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga, (double)si[g * 16 * PITCH + 4 * s4], acci[g], 0, 0, 0);
            }
        }
        wave_lds_sync();
    }
    __syncthreads();  // the scan's exchange array lies over every wave's rows

    // ---- S / L: per component: chunk states to one thread per chunk, from-rest scan, publish the segment's end state ----
    double z0[D], z1[D];
    auto scan_component = [&](const v4d_t (&acc)[4], double (&z)[D], int comp) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = j + 4 * r;
                if (d < D) sc[d * kIirThreads + wave * 64 + g * 16 + c] = acc[g][r];
            }
        __syncthreads();
        double v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = sc[d * kIirThreads + tid];
        __syncthreads();
#pragma unroll 1
        for (int l = 0; l < a.n_lv; ++l) {
            const int s = 1 << l;
#pragma unroll
            for (int d = 0; d < D; ++d) sc[d * kIirThreads + tid] = v[d];
            __syncthreads();
            if (tid >= s) {
                double left[D];
#pragma unroll
                for (int d = 0; d < D; ++d) left[d] = sc[d * kIirThreads + tid - s];
                matvec_acc<D, ORD>(pw + (size_t)l * D * D, left, v);
            }
            __syncthreads();
        }
#pragma unroll
        for (int d = 0; d < D; ++d) sc[d * kIirThreads + tid] = v[d];
        __syncthreads();
#pragma unroll
        for (int d = 0; d < D; ++d) z[d] = tid ? sc[d * kIirThreads + tid - 1] : 0.0;
        if (tid < 2 * D) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(sc[(tid >> 1) * kIirThreads + kIirThreads - 1]);
            const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
            __hip_atomic_store(a.lb + (size_t)seg * 64 + comp * 32 + tid, ((unsigned long long)a.epoch << 32) | half, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    };
    scan_component(accr, z0, 0);
    scan_component(acci, z1, 1);
    if (tid < 4 * D) {   // lanes [0, 2 D): re, [2 D, 4 D): im
        const int comp = tid >= 2 * D, q = tid - comp * 2 * D;
        unsigned got = 0;
        if (seg > 0) {
            const unsigned long long *src = a.lb + (size_t)(seg - 1) * 64 + comp * 32 + q;
            unsigned long long g = 0;
            int spins = 0;
            for (;;) {
                g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(g >> 32) == a.epoch) break;
                if (++spins > (1 << 22)) {
                    *a.err = 1u;
                    g = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            got = (unsigned)g;
        } else if (a.zi) {
            const unsigned long long zb = (unsigned long long)__double_as_longlong(a.zi[(size_t)comp * D + (q >> 1)]);
            got = (q & 1) ? (unsigned)(zb >> 32) : (unsigned)zb;
        }
        cw[comp * 32 + q] = got;
    }
    __syncthreads();

    // ---- C: z_j += M^j c, per component ----
    if ((wave << 6) < (1 << a.n_lv)) {
#pragma unroll 1
        for (int comp = 0; comp < 2; ++comp) {
            double u[D];
#pragma unroll
            for (int d = 0; d < D; ++d)
                u[d] = __longlong_as_double((long long)(((unsigned long long)cw[comp * 32 + 2 * d + 1] << 32) | cw[comp * 32 + 2 * d]));
#pragma unroll 1
            for (int l = 0; l < a.n_lv; ++l) {
                if ((tid >> l) & 1) {
                    double t2[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) t2[d] = 0.0;
                    matvec_acc<D, ORD>(pw + (size_t)l * D * D, u, t2);
#pragma unroll
                    for (int d = 0; d < D; ++d) u[d] = t2[d];
                }
            }
            if (tid < (1 << a.n_lv)) {
                if (comp == 0) {
#pragma unroll
                    for (int d = 0; d < D; ++d) z0[d] += u[d];
                } else {
#pragma unroll
                    for (int d = 0; d < D; ++d) z1[d] += u[d];
                }
            }
        }
    }

    // ---- B: both recurrences over the register-resident chunk; outputs re-joined through the planes ----
    const int64_t cj = row0 + tid;
    const bool zf_owner = a.zf != nullptr && cj == (a.n - 1) / TC;
    const int zf_off = (int)((a.n - 1) % TC);
    const int zf_piece = (a.zf != nullptr && (a.n - 1) / TC >= row0 && (a.n - 1) / TC < row0 + kIirThreads) ? zf_off / PC : -1;  // uniform
    __syncthreads();  // (the planes are free again: every wave has read its scan results)
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
        if (p == zf_piece && zf_owner) {  // (see iir_fused_kernel)
            double zt0[D], zt1[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { zt0[d] = z0[d]; zt1[d] = z1[d]; }
            const IO *xp = x + 2 * (cj * TC + (int64_t)p * PC);
#pragma unroll 1
            for (int k = 0; k <= zf_off - p * PC; ++k) {
                (void)cascade_step<NSEC, ORD, UNIT>(cf, zt0, (double)xp[2 * k]);
                (void)cascade_step<NSEC, ORD, UNIT>(cf, zt1, (double)xp[2 * k + 1]);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) { a.zf[d] = zt0[d]; a.zf[D + d] = zt1[d]; }
        }
#pragma unroll
        for (int k = 0; k < PC; ++k) xr[k] = (IO)cascade_step<NSEC, ORD, UNIT>(cf, z0, (double)xr[k]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < PC; ++k) xi[k] = (IO)cascade_step<NSEC, ORD, UNIT>(cf, z1, (double)xi[k]);
        wave_lds_sync();
#pragma unroll
        for (int c4 = 0; c4 < PC / E; ++c4) {
            float4 qa, qb;
            IO *ea = reinterpret_cast<IO *>(&qa), *eb = reinterpret_cast<IO *>(&qb);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ea[e] = xr[c4 * E + e];
                eb[e] = xi[c4 * E + e];
            }
            *reinterpret_cast<float4 *>(rr + c4 * E) = qa;
            *reinterpret_cast<float4 *>(ri + c4 * E) = qb;
        }
#pragma unroll
        for (int k = 0; k + PC < TC; ++k) {
            xr[k] = xr[k + PC];
            xi[k] = xi[k + PC];
        }
        wave_lds_sync();
        int64_t dq_run = 0;
        int dr_run = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = i * 64 + lane;
            const int row = wave * 64 + (idx >> 3), sg = idx & 7;
            const int64_t g = (row0 + row) * TC + (int64_t)p * PC + (int64_t)sg * (E / 2);
            IO out[E];
#pragma unroll
            for (int k = 0; k < E / 2; ++k) {
                out[2 * k] = st_re[row * PITCH + sg * (E / 2) + k];
                out[2 * k + 1] = st_im[row * PITCH + sg * (E / 2) + k];
            }
            if (a.dec > 1) {   // at most one of the E / 2 <= 2 <= dec complex samples of a segment is kept
                if (i == 0) {
                    dq_run = g / a.dec;
                    dr_run = (int)(g - dq_run * a.dec);
                }
                const int e0 = dr_run == 0 ? 0 : a.dec - dr_run;
                if (e0 < E / 2 && g + e0 < a.n_keep) {
                    IO re = out[0], im = out[1];
#pragma unroll
                    for (int k = 1; k < E / 2; ++k) {
                        re = (e0 == k) ? out[2 * k] : re;
                        im = (e0 == k) ? out[2 * k + 1] : im;
                    }
                    IO *dst = y + 2 * (dq_run + (dr_run != 0));
                    dst[0] = re;
                    dst[1] = im;
                }
                dq_run += a.dec_dq;
                dr_run += a.dec_dr;
                if (dr_run >= a.dec) { dr_run -= a.dec; ++dq_run; }
            } else if (interior || g + E / 2 <= a.n) {
                __builtin_nontemporal_store(*reinterpret_cast<const pre_t *>(out), reinterpret_cast<pre_t *>(y + 2 * g));
            } else if (g < a.n) {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (g + e / 2 < a.n) y[2 * g + e] = out[e];
            }
        }
    }
}

// single-pass scan: one workgroup per segment of 256 chunks, segments handed out by ticket
template <typename IO>
static int launch_fused(IirHandle *h, const void *x, int64_t n, int nbatch, int64_t batch_stride, void *y, const double *zi_dev,
                        double *zf_dev, hipStream_t s, int dec, int interleaved)
{
    IirPlan *p = h->plan;
    const int T = (interleaved ? 256 : 512) / (int)sizeof(IO);
    const int64_t S = (int64_t)kIirThreads * T;
    const int64_t nseg = (n + S - 1) / S;
    SK_CHECK(nseg < (1 << 30), SKDSP_ERR_BADARG, "iir: too many segments");
    if (!p->ticket_dev) {
        SK_HIP(hipMalloc((void **)&p->ticket_dev, 16));
        SK_HIP(hipMemsetAsync(p->ticket_dev, 0, 16, s));
        p->ticket_count[0] = p->ticket_count[1] = 0;
    }
    const size_t need = (size_t)(interleaved ? 2 : nbatch) * nseg * 32 * 8;
    if (need > p->lbg_cap) {
        if (p->lbg_dev) {
            SK_HIP(hipStreamSynchronize(s));
            SK_HIP(hipFree(p->lbg_dev));
        }
        p->lbg_dev = nullptr; p->lbg_cap = 0;
        SK_HIP(hipMalloc((void **)&p->lbg_dev, need));
        SK_HIP(hipMemsetAsync(p->lbg_dev, 0, need, s));  // epoch 0 is never used as a tag
        p->lbg_cap = need;
    }
    FusedArgs a;
    a.x = x; a.y = y; a.n = n; a.batch_stride = batch_stride;
    a.pw = p->pw_dev; a.gt = p->gt_dev;
    a.lb = p->lbg_dev; a.ticket = p->ticket_dev;
    a.ticket_base[0] = p->ticket_count[0]; a.ticket_base[1] = p->ticket_count[1];
    a.epoch = ++p->epoch;
    if (a.epoch == 0) a.epoch = ++p->epoch;
    a.nseg = (int)nseg; a.n_lv = p->n_lv < 8 ? p->n_lv : 8;
    a.zi = zi_dev; a.zf = zf_dev;
    a.dec = dec > 1 ? dec : 1;
    a.n_keep = (n / a.dec) * a.dec;
    {
        // samples between a thread's staged segments: a wave stages 64 / segs rows per step (8 segments per row piece in the
        // interleaved kernel)
        const int64_t step = (int64_t)(interleaved ? 64 / 8 : 64 / Stage<IO>::segs) * T;
        a.dec_dq = (int)(step / a.dec);
        a.dec_dr = (int)(step % a.dec);
    }
    // a poll that gives up is reported (and cleared) by the call that next synchronises this slot's stream (async_err_check):
    // every workgroup has drawn its ticket before it polls and the granules carry the launch's epoch, so the dispenser and
    // the look-back array stay consistent and later calls on this handle are unaffected
    a.err = async_err_dev(kAsyncErrIirLookback);
    SK_CHECK(a.err, SKDSP_ERR_HIP, "iir: no host-mapped error word");
    for (int b = 0; b < (interleaved ? 1 : nbatch); ++b) p->ticket_count[b] += (unsigned long long)nseg;
    const dim3 grid((unsigned)nseg, (unsigned)(interleaved ? 1 : nbatch));
    if (interleaved) {
#define SK_FUSEDC(N)                                                                                                  \
    case N: {                                                                                                        \
        Coef<N, 2> cf;                                                                                               \
        std::memcpy(cf.c, h->coef.data(), sizeof(cf.c));                                                             \
        if (N >= 2 && h->unit_tail) hipLaunchKernelGGL((iir_fused_c_kernel<N, IO, (N >= 2)>), grid, dim3(kIirThreads), 0, s, a, cf, a.pw, a.gt); \
        else hipLaunchKernelGGL((iir_fused_c_kernel<N, IO, false>), grid, dim3(kIirThreads), 0, s, a, cf, a.pw, a.gt); \
        break;                                                                                                       \
    }
        switch (h->nsec) {
            SK_FUSEDC(1) SK_FUSEDC(2) SK_FUSEDC(3) SK_FUSEDC(4) SK_FUSEDC(5) SK_FUSEDC(6) SK_FUSEDC(7)
            SK_FUSEDC(8)
            default: SK_CHECK(false, SKDSP_ERR_UNSUPPORTED, "iir: single-pass scan takes 1..8 biquads");
        }
#undef SK_FUSEDC
        SK_HIP(hipGetLastError());
        return SKDSP_OK;
    }
#define SK_FUSED(N)                                                                                                  \
    case N: {                                                                                                        \
        Coef<N, 2> cf;                                                                                               \
        std::memcpy(cf.c, h->coef.data(), sizeof(cf.c));                                                             \
        if (N >= 2 && h->unit_tail) hipLaunchKernelGGL((iir_fused_kernel<N, IO, (N >= 2)>), grid, dim3(kIirThreads), 0, s, a, cf, a.pw, a.gt, (const double *)p->pwa_dev); \
        else hipLaunchKernelGGL((iir_fused_kernel<N, IO, false>), grid, dim3(kIirThreads), 0, s, a, cf, a.pw, a.gt, (const double *)p->pwa_dev);  \
        break;                                                                                                       \
    }
    switch (h->nsec) {
        SK_FUSED(1) SK_FUSED(2) SK_FUSED(3) SK_FUSED(4) SK_FUSED(5) SK_FUSED(6) SK_FUSED(7)
        SK_FUSED(8)
        default: SK_CHECK(false, SKDSP_ERR_UNSUPPORTED, "iir: single-pass scan takes 1..8 biquads");
    }
#undef SK_FUSED
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

int iir_fused_launch(IirHandle *h, const void *x, int64_t n, int nbatch, int64_t batch_stride, void *y, const double *zi_dev,
                     double *zf_dev, hipStream_t s, int dec, int interleaved)
{
    note_path("iir_fused");
    return dtype_double(h->dtype) ? launch_fused<double>(h, x, n, nbatch, batch_stride, y, zi_dev, zf_dev, s, dec, interleaved)
                                  : launch_fused<float>(h, x, n, nbatch, batch_stride, y, zi_dev, zf_dev, s, dec, interleaved);
}

}  // namespace skdsp
