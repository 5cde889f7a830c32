// fir_ols.hip -- LDS/register-resident FFT overlap-save FIR for gfx950 (MI355X).
//
// Serves multirate_FIR.filter (multirate_helper.py:104-109, lfilter(b,[1],x)) for
// complex64 signals with long filters -- the headline config (1024 taps, 2^26
// samples).  Direct form needs 4*P flop per sample (P=1024: 8 % of the HBM roofline
// at FP32 peak); in the frequency domain the same outputs cost
// ~120 flop per sample, so the kernel can be HBM-bound.
//
// One workgroup (256 threads, 4 waves) works on one tile of N = 8192 points at a time
// (persistent: 2 workgroups per CU walk all tiles; compiled with -fno-slp-vectorize):
//   load  x[tile*V - OV .. +8192)  straight into registers, 16 x 16-byte loads/lane,
//         issued one tile ahead so the HBM latency hides under the previous FFT
//   forward FFT (DIF, 16 x 16 x 32, see ols_core.hpp) -> multiply by H -> inverse FFT (DIT)
//   store the last V = 8192 - OV points (OV = overlap, a multiple of 512 >= P-1)
// The only HBM traffic is the input tile (read once, + OV/V overlap re-read that the
// 256 MiB MALL absorbs) and the V outputs; twiddles (64 KiB + 4 KiB) and H (64 KiB)
// are L2-resident tables.  Algorithmic bytes = 16 B per sample (8 in + 8 out).
//
// Precision: float32 butterflies with float64-derived constants; measured 2.4e-7
// max-abs/peak against a float64 direct FIR (tests/host/ols_emul.cpp, GPU parity tests).
#include "skdsp_internal.hpp"
#include "ols_tables.hpp"
#include <cstdlib>
#include <cstdio>


namespace skdsp {

using namespace ols;
typedef float v4f_t __attribute__((ext_vector_type(4)));

struct OlsPlan {
    int ntaps = 0;
    int ov = 0;  // overlap (discarded head) in samples, multiple of 512
    int V = 0;   // valid outputs per tile
    float4 *T1 = nullptr, *T2 = nullptr, *Hp = nullptr;
};

struct OlsArgs {
    const cf *x;
    cf *y;
    int64_t n, n_hist;
    const float4 *T1, *T2, *Hp;
    int ov, V, a0;  // a0 = ov / 512: first stored 512-block
    // The first a0 and the last a0 512-sample blocks of a tile are what it shares with its neighbours (the overlap it reads from the previous tile's range, and
    // the range the next tile will read as ITS overlap).  keep = a0: those blocks are requested with ordinary loads, so that the lines stay in the XCD's L2 for the
    // neighbour (a nontemporal line is the first to be evicted); the blocks in between stay nontemporal.  0: every load nontemporal (round 5).
    int keep;
    int aligned;    // x and y element-aligned (8 bytes complex64, 4 bytes float32)
    int64_t ntiles;
    int dec;        // > 1: keep every dec-th output only (multirate_FIR.dn): y[g / dec] = out[g] for g % dec == 0
    unsigned dec_magic;   // ceil(2^32 / dec): (g * dec_magic) >> 32 = g / dec for the tile-local g < 2^15 met in the store
    // up > 1 (multirate_FIR.up with long phases): the walk runs over (tile, phase) pairs -- index w stands for input tile w / up filtered
    // with the taps of phase w % up (Hp holds up tables of 4096 float4), and output i of that pair lands at y[i * up + phase] --
    // or, up_pitch > 0, at y[phase * up_pitch + i]: the phases as rows (scratch), woven together by interleave_launch afterwards
    int up;
    int64_t up_pitch;
    // the strided store of a pair: output i of local phase q lands up_sb * i + up_pb0 + up_pbs * q BYTES behind y (the plain cases: up_sb =
    // up * element size, up_pbs = element size, up_pb0 = 0; a launch may also cover SOME phases of an L-fold interpolation -- pairs of
    // phases as 8-byte elements in a float32 output, or the last phase of an odd L on its own)
    int up_sb, up_pbs, up_pb0;
    int64_t n_keep; // dec * floor(n / dec)
    // Sharded filter (dist.hip): the Ntaps-1 samples in front of x arrive over xGMI on another stream while this launch
    // already runs.  Only tile 0 reads them, so tile 0 is walked LAST and whoever owns it waits for halo_flag >= halo_seq
    // (set by a one-thread kernel behind the RCCL receive) right before requesting its samples.  null = no wait.
    const unsigned *halo_flag;
    unsigned halo_seq;
    unsigned *halo_err;  // host-mapped: set if the bounded wait gave up (the caller reports it; the launch never hangs)
    int halo_spins;      // polls (~1 us each) before giving up: ~2^21 = seconds in steady state, a few thousand in the probation step of dist.hip
    CarefulFir cf;       // the filter as the exact path of a poisoned tile reads it (careful.hpp)
    // ols_rep_kernel (multirate_FIR.up on the replicated spectrum): x holds n_in samples at the LOW rate (rep_hist of history in front), n = n_in L;
    // rep_lr = L / LF, what is left of L beyond the power of two LF the kernel is compiled for (the zero-stuffed grid is itself stuffed rep_lr-fold)
    int rep_L, rep_lr;
    unsigned rep_magic;  // ceil(2^32 / rep_lr)
    int64_t n_in, rep_hist;
};

// The owner of tile 0 calls this (whole workgroup, uniform) before its first load of that tile: one lane polls with
// relaxed agent-scope loads, then ONE agent-scope acquire (invalidates this CU's L1; the samples were written by another
// kernel / another GPU), then the barrier releases the other waves to plain loads (MI355X_MICROARCH.md, inter-workgroup
// visibility).  The halo left its sender ~0.2 ms earlier, so the loop normally exits on its first load.
__device__ __forceinline__ void wait_halo(const OlsArgs &A)
{
    if (threadIdx.x == 0) {
        int spins = 0;
        while ((int)(__hip_atomic_load(A.halo_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.halo_seq) < 0) {
            if (++spins > A.halo_spins) {  // report instead of hanging the GPU
                *A.halo_err = 1u;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// volatile 16-byte load: keeps the request at its program position (the scheduler would
// otherwise sink a prefetch down to its first use to save registers)
__device__ __forceinline__ float4 vld(const volatile float4 *p)
{
    float4 r;
    r.x = p->x; r.y = p->y; r.z = p->z; r.w = p->w;
    return r;
}

// x[in0 + 512 a + 2 t + e] -> v[2a+e]; zero outside [-n_hist, n)
__device__ __forceinline__ void load_tile(const OlsArgs &A, int64_t tile, int t, cf *v)
{
    const int64_t in0 = tile * A.V - A.ov;
    const bool interior = A.aligned && in0 >= -A.n_hist && in0 + kN <= A.n;
    if (interior) {
        // opaque copy of t: stops LICM from hoisting 16 loop-invariant 64-bit addresses (which
        // were then spilled and reloaded in front of every load)
        int tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const v4f_t *src = reinterpret_cast<const v4f_t *>(A.x + in0) + (unsigned)(a * 256 + tt);
            v4f_t nv;
            if (a < A.keep || a >= 16 - A.keep) nv = *src;   // (wave-uniform: the blocks a neighbouring tile reads too)
            else nv = __builtin_nontemporal_load(src);
            const float4 f = make_float4(nv.x, nv.y, nv.z, nv.w);
            v[2 * a] = lo(f);
            v[2 * a + 1] = hi(f);
        }
    } else {
        int tt = t;   // (opaque copy: the 64-bit lane offset of this edge-tile path is not kept -- spilled -- across the tile loop)
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int a = 0; a < 16; ++a) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int64_t g = in0 + 512 * a + 2 * tt + e;
                cf val = make_float2(0.f, 0.f);
                if (g >= -A.n_hist && g < A.n) val = A.x[g];
                v[2 * a + e] = val;
            }
        }
    }
}

typedef float v2f_t __attribute__((ext_vector_type(2)));

template <bool DEC> __device__ __forceinline__ void store_tile(const OlsArgs &A, int64_t tile, int t, const cf *v, float4 *lds)
{
    const int64_t out0 = tile * A.V;
    const bool full = A.aligned && out0 + A.V <= A.n;
    if (DEC) {
        // decimating store: the full-rate convolution is computed (it is memory-bound, and cheaper than
        // Ntaps/dec direct taps per kept sample once Ntaps/dec exceeds a few dozen), 1/dec of it leaves.  The tile's kept outputs are
        // ONE run of y: every thread drops its kept samples into the (idle) FFT image at their output positions -- one multiply-high
        // per sample finds them -- and the workgroup then writes the run with consecutive 2 KiB stores.  (Before: a 32-bit division
        // and a predicated 8-byte store per sample, a quarter of the lanes of every store instruction active at dec = 4.)
        const unsigned M = (unsigned)A.dec;
        const int64_t q0 = out0 / A.dec;                 // uniform
        const unsigned r0 = (unsigned)(out0 - q0 * A.dec);
        const unsigned ob = r0 != 0 ? 1u : 0u;           // the tile's first output, relative to q0
        cf *buf = reinterpret_cast<cf *>(lds);
        int a0 = A.a0;   // (opaque copy: nothing of this path is hoisted out of the tile loop)
        asm volatile("" : "+s"(a0));
        __syncthreads();   // every wave has read its share of the image
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            if (a < a0) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const unsigned gl = r0 + 512u * (unsigned)(a - a0) + 2u * (unsigned)t + (unsigned)e;
                const unsigned q = (unsigned)(((unsigned long long)gl * A.dec_magic) >> 32);
                if (q * M == gl) buf[q - ob] = v[2 * a + e];
            }
        }
        __syncthreads();
        int64_t oe = q0 + (int64_t)((r0 + (unsigned)A.V + M - 1) / M);
        const int64_t n_out = A.n_keep / A.dec;
        if (oe > n_out) oe = n_out;
        const int cnt = (int)(oe - (q0 + ob));
        cf *yo = A.y + q0 + ob;
        for (int i = t; i < cnt; i += 256) {
            const v2f_t o = {buf[i].x, buf[i].y};
            __builtin_nontemporal_store(o, reinterpret_cast<v2f_t *>(yo + i));
        }
        return;
    }
    if (full) {
        // recompute the per-thread offset here: hoisted out of the tile loop it is a 64-bit VGPR pair
        // that hipcc spills, and the scratch reload's s_waitcnt vmcnt(0) then drains the whole
        // x(tile+1) prefetch in front of the stores (vmcnt retires in order)
        int tt = t;
        asm volatile("" : "+v"(tt));
        float4 *yp = reinterpret_cast<float4 *>(A.y + out0 + 2 * tt);
        // one copy of the 16 - a0 stores per possible a0 (compile-time offsets and no predicates): with a run-time a0 hipcc
        // hoisted sixteen (64-bit exec mask, 64-bit offset) pairs out of the tile loop -- 96 SGPRs, parked in VGPR lanes
        // and fetched back with 139 v_readlane per tile
        auto stores = [&](auto a0c) __attribute__((always_inline)) {
            constexpr int A0 = decltype(a0c)::value;
#pragma unroll
            for (int a = A0; a < 16; ++a) {
                v4f_t nv;
                nv.x = v[2 * a].x; nv.y = v[2 * a].y; nv.z = v[2 * a + 1].x; nv.w = v[2 * a + 1].y;
                __builtin_nontemporal_store(nv, reinterpret_cast<v4f_t *>(yp) + (a - A0) * 256);
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
        int a0 = A.a0;   // opaque copy: keeps the sixteen (mask, offset) pairs of this once-per-launch path out of the tile loop's prologue
        asm volatile("" : "+s"(a0));
        int tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            if (a < a0) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int64_t g = out0 + 512 * (a - a0) + 2 * tt + e;
                if (g < A.n) A.y[g] = v[2 * a + e];
            }
        }
    }
}

// ---- float32 signals with real taps: two real tiles ride in one complex tile ----------
// A real-tap FIR commutes with taking real/imaginary parts, so real tile 2p goes in as
// the real part and real tile 2p+1 as the imaginary part of complex tile p; the FFT work
// per real sample halves and the same kernel body serves both dtypes.

__device__ __forceinline__ void load_tile_real(const OlsArgs &A, int64_t pair, int t, cf *v)
{
    const float *xr = reinterpret_cast<const float *>(A.x);
    const int64_t inA = (2 * pair) * A.V - A.ov, inB = inA + A.V;
    const bool interior = A.aligned && inA >= -A.n_hist && inB + kN <= A.n;
    if (interior) {
        int tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            // v[2a] = the (column 0, column 1) pair of tile A, v[2a+1] of tile B: exactly what fwd_pass1_real takes
            // (structure of arrays; ols_core.hpp) -- the loaded registers are used where they land
            const v2f_t ra = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(xr + inA) + (unsigned)(a * 256 + tt));
            const v2f_t rb = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(xr + inB) + (unsigned)(a * 256 + tt));
            v[2 * a] = make_float2(ra.x, ra.y);
            v[2 * a + 1] = make_float2(rb.x, rb.y);
        }
    } else {
        int tt = t;   // (opaque copy: see load_tile)
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            float ra[2], rb[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int64_t ga = inA + 512 * a + 2 * tt + e, gb = ga + A.V;
                ra[e] = (ga >= -A.n_hist && ga < A.n) ? xr[ga] : 0.f;
                rb[e] = (gb >= -A.n_hist && gb < A.n) ? xr[gb] : 0.f;
            }
            v[2 * a] = make_float2(ra[0], ra[1]);
            v[2 * a + 1] = make_float2(rb[0], rb[1]);
        }
    }
}

template <bool DEC> __device__ __forceinline__ void store_tile_real(const OlsArgs &A, int64_t pair, int t, const cf *v, float4 *lds)
{
    float *yr = reinterpret_cast<float *>(A.y);
    const int64_t outA = (2 * pair) * A.V, outB = outA + A.V;
    const bool full = A.aligned && outB + A.V <= A.n;
    if (DEC) {  // decimating store, see store_tile: the pair's two tiles are one run of 2 V outputs
        const unsigned M = (unsigned)A.dec;
        const int64_t q0 = outA / A.dec;
        const unsigned r0 = (unsigned)(outA - q0 * A.dec);
        const unsigned ob = r0 != 0 ? 1u : 0u;
        float *buf = reinterpret_cast<float *>(lds);
        int a0 = A.a0;   // (opaque copy: see store_tile)
        asm volatile("" : "+s"(a0));
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            if (a < a0) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const unsigned ga = r0 + 512u * (unsigned)(a - a0) + 2u * (unsigned)t + (unsigned)e, gb = ga + (unsigned)A.V;
                const unsigned ka = (unsigned)(((unsigned long long)ga * A.dec_magic) >> 32), kb = (unsigned)(((unsigned long long)gb * A.dec_magic) >> 32);
                if (ka * M == ga) buf[ka - ob] = v[2 * a + e].x;
                if (kb * M == gb) buf[kb - ob] = v[2 * a + e].y;
            }
        }
        __syncthreads();
        int64_t oe = q0 + (int64_t)((r0 + 2u * (unsigned)A.V + M - 1) / M);
        const int64_t n_out = A.n_keep / A.dec;
        if (oe > n_out) oe = n_out;
        const int cnt = (int)(oe - (q0 + ob));
        float *yo = yr + q0 + ob;
        for (int i = t; i < cnt; i += 256) __builtin_nontemporal_store(buf[i], yo + i);
        return;
    }
    if (full) {
        int tt = t;  // (opaque copy: keeps the 28 store addresses from being hoisted out of the tile loop and spilled)
        asm volatile("" : "+v"(tt));
        auto stores = [&](auto a0c) __attribute__((always_inline)) {   // (one copy per a0: see store_tile)
            constexpr int A0 = decltype(a0c)::value;
#pragma unroll
            for (int a = A0; a < 16; ++a) {
                // v[2a] = (A, B) of column 0, v[2a+1] = (A, B) of column 1: ONE register swap turns the two pairs into
                // (A col 0, A col 1) and (B col 0, B col 1), the 8-byte store operands
                float a0 = v[2 * a].x, b0 = v[2 * a].y, a1 = v[2 * a + 1].x, b1 = v[2 * a + 1].y;
                asm("v_swap_b32 %0, %1" : "+v"(b0), "+v"(a1));
                v2f_t ra, rb;
                ra.x = a0; ra.y = b0;   // (b0 now holds A of column 1)
                rb.x = a1; rb.y = b1;   // (a1 now holds B of column 0)
                __builtin_nontemporal_store(ra, reinterpret_cast<v2f_t *>(yr + outA + 2 * tt) + (a - A0) * 256);
                __builtin_nontemporal_store(rb, reinterpret_cast<v2f_t *>(yr + outB + 2 * tt) + (a - A0) * 256);
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
        int a0 = A.a0;   // (opaque copy: see store_tile)
        asm volatile("" : "+s"(a0));
        int tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            if (a < a0) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int64_t ga = outA + 512 * (a - a0) + 2 * tt + e, gb = ga + A.V;
                if (ga < A.n) yr[ga] = v[2 * a + e].x;
                if (gb < A.n) yr[gb] = v[2 * a + e].y;
            }
        }
    }
}

// A REAL signal into the COMPLEX tile (imaginary part zero): multirate_FIR.up of float32 signals with an even L runs its phases in
// pairs -- x * (h_2k + i h_2k+1) = y_2k + i y_2k+1 is one complex pass, and its output IS the interleaved pair (y[iL + 2k], y[iL + 2k + 1])
// as one 8-byte element.  Same number of transforms as two real tiles per complex tile and one phase per pass, but the stores are 8 bytes
// wide -- contiguous at L = 2 -- and the input is read once per pair.
__device__ __forceinline__ void load_tile_xr(const OlsArgs &A, int64_t tile, int t, cf *v)
{
    const float *xr = reinterpret_cast<const float *>(A.x);
    const int64_t in0 = tile * A.V - A.ov;
    const bool interior = A.aligned && in0 >= -A.n_hist && in0 + kN <= A.n;
    int tt = t;
    asm volatile("" : "+v"(tt));
    if (interior) {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const v2f_t r = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(xr + in0) + (unsigned)(a * 256 + tt));
            v[2 * a] = make_float2(r.x, 0.f);
            v[2 * a + 1] = make_float2(r.y, 0.f);
        }
    } else {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int64_t g = in0 + 512 * a + 2 * tt + e;
                v[2 * a + e] = make_float2((g >= -A.n_hist && g < A.n) ? xr[g] : 0.f, 0.f);
            }
        }
    }
}

template <bool REAL, bool XR = false> __device__ __forceinline__ void load_any(const OlsArgs &A, int64_t tile, int t, cf *v)
{
    if (XR) load_tile_xr(A, tile, t, v); else if (REAL) load_tile_real(A, tile, t, v); else load_tile(A, tile, t, v);
}
template <bool REAL, bool DEC> __device__ __forceinline__ void store_any(const OlsArgs &A, int64_t tile, int t, const cf *v, float4 *lds)
{
    if (REAL) store_tile_real<DEC>(A, tile, t, v, lds); else store_tile<DEC>(A, tile, t, v, lds);
}

// ---- multirate_FIR.up: output i of (tile, phase) goes to y[i * up + phase] -------------------------------------------
// Every phase touches every 128-byte line of the tile's output run, whatever the lanes do; what can be chosen is how many lines ONE
// store instruction touches.  A thread holds outputs 2t and 2t+1 of each 512-block, so storing them as they lie makes an instruction
// span 128 outputs with every other one written.  One v_permlane32_swap per register first hands the odd outputs of lanes 0..31 to
// lanes 32..63 and the even outputs of lanes 32..63 to lanes 0..31: each instruction then writes 64 CONSECUTIVE outputs of the
// phase (half the lines per instruction, both for 8-byte and 4-byte samples).
__device__ __forceinline__ void lanes_swap_halves(float &e0, float &e1)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
    e0 = __uint_as_float(r[0]);   // lanes 0..31: own even output; lanes 32..63: odd output of lane - 32
    e1 = __uint_as_float(r[1]);   // lanes 0..31: even output of lane + 32; lanes 32..63: own odd output
}
// position (inside a 512-block) of the first of this lane's two outputs after the swap; the second lies 64 further
__device__ __forceinline__ int up_lane_pos(int t)
{
    const int lane = t & 63;
    return 128 * (t >> 6) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);
}

// (addresses: a uniform 64-bit base per 512-block plus a 32-bit per-lane byte offset -- the scalar-base form of the store instruction;
// per-lane 64-bit addresses for the 32 stores were all formed ahead of the first store and cost 14-62 spilled registers)
template <bool DEC> __device__ __forceinline__ void store_tile_up(const OlsArgs &A, int64_t tile, int ph, int t, const cf *v)
{
    int a0 = A.a0;
    asm volatile("" : "+s"(a0));
    int tt = t;   // (opaque copy: the offsets are rebuilt per tile instead of living in registers across the tile loop)
    asm volatile("" : "+v"(tt));
    const int pos = up_lane_pos(tt);   // this lane's first output inside a 512-block (the second: + 64)
    const int64_t out0 = tile * A.V;
    char *ub = reinterpret_cast<char *>(A.y) + out0 * A.up_sb + A.up_pb0 + (int64_t)ph * A.up_pbs;   // uniform
    const int64_t left = A.n - out0;
    const int lim = (left > (1 << 20) ? (1 << 20) : (int)left) - pos;   // outputs i < lim (relative to this lane's first) exist
    const bool whole = left >= A.V;   // (uniform: every tile but the last)
    const unsigned b0 = (unsigned)pos * (unsigned)A.up_sb, b1 = b0 + 64u * (unsigned)A.up_sb;
    const size_t step = (size_t)512 * A.up_sb;
    // DEC (L / M): the tile's first up-rate index out0 up + ph = q0 M + r0; tile-local up-rate indices stay below 2^20 + M, where the
    // multiply-high by ceil(2^32 / M) is the exact quotient for M <= 4096 (checked at launch); n_keep = floor(n up / M) outputs exist
    const unsigned M = (unsigned)A.dec;
    const int64_t jt = out0 * A.up + ph, q0 = DEC ? jt / A.dec : 0;
    const unsigned r0 = DEC ? (unsigned)(jt - q0 * A.dec) : 0u;
    const int64_t qleft = A.n_keep - q0;
    const int qlim = qleft > (1 << 24) ? (1 << 24) : (int)qleft;
    cf *yq = A.y + q0;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        if (a < a0) continue;
        cf e0 = v[2 * a], e1 = v[2 * a + 1];
        lanes_swap_halves(e0.x, e1.x);
        lanes_swap_halves(e0.y, e1.y);
        const int i = 512 * (a - a0);
        if (DEC) {   // L / M: up-rate index j = (out0 + i) up + ph is kept iff M divides it, at y[j / M]
            const unsigned j0 = r0 + (unsigned)(i + pos) * (unsigned)A.up, j1 = j0 + 64u * (unsigned)A.up;
            const unsigned k0 = (unsigned)(((unsigned long long)j0 * A.dec_magic) >> 32), k1 = (unsigned)(((unsigned long long)j1 * A.dec_magic) >> 32);
            if (k0 * M == j0 && (int)k0 < qlim && (whole || i < lim)) yq[k0] = e0;
            if (k1 * M == j1 && (int)k1 < qlim && (whole || i + 64 < lim)) yq[k1] = e1;
            continue;
        }
        char *ua = ub + (size_t)(a - a0) * step;
        if (whole || i < lim) *reinterpret_cast<cf *>(ua + b0) = e0;
        if (whole || i + 64 < lim) *reinterpret_cast<cf *>(ua + b1) = e1;
    }
}

template <bool DEC> __device__ __forceinline__ void store_tile_real_up(const OlsArgs &A, int64_t pair, int ph, int t, const cf *v)
{
    int a0 = A.a0;
    asm volatile("" : "+s"(a0));
    int tt = t;
    asm volatile("" : "+v"(tt));
    const int pos = up_lane_pos(tt);
    const int64_t outA = (2 * pair) * A.V;
    char *ua0 = reinterpret_cast<char *>(A.y) + outA * A.up_sb + A.up_pb0 + (int64_t)ph * A.up_pbs;   // uniform
    char *ub0 = ua0 + (size_t)A.V * A.up_sb;   // the pair's second tile
    const int64_t left = A.n - outA;
    const int lim = (left > (1 << 20) ? (1 << 20) : (int)left) - pos;
    const int limb = lim - A.V;
    const bool whole = left >= 2 * (int64_t)A.V;
    const unsigned b0 = (unsigned)pos * (unsigned)A.up_sb, b1 = b0 + 64u * (unsigned)A.up_sb;
    const size_t step = (size_t)512 * A.up_sb;
    const unsigned M = (unsigned)A.dec;   // (DEC: see store_tile_up)
    const int64_t jt = outA * A.up + ph, q0 = DEC ? jt / A.dec : 0;
    const unsigned r0 = DEC ? (unsigned)(jt - q0 * A.dec) : 0u;
    const int64_t qleft = A.n_keep - q0;
    const int qlim = qleft > (1 << 24) ? (1 << 24) : (int)qleft;
    float *yq = reinterpret_cast<float *>(A.y) + q0;
    const unsigned jv = (unsigned)A.V * (unsigned)A.up;   // the second tile's offset at the up rate
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        if (a < a0) continue;
        cf e0 = v[2 * a], e1 = v[2 * a + 1];
        lanes_swap_halves(e0.x, e1.x);
        lanes_swap_halves(e0.y, e1.y);
        const int i = 512 * (a - a0);
        if (DEC) {
            const unsigned j0 = r0 + (unsigned)(i + pos) * (unsigned)A.up, j1 = j0 + 64u * (unsigned)A.up;
            auto keep = [&](unsigned j, float val, bool inside) __attribute__((always_inline)) {
                const unsigned k = (unsigned)(((unsigned long long)j * A.dec_magic) >> 32);
                if (k * M == j && (int)k < qlim && inside) yq[k] = val;
            };
            keep(j0, e0.x, whole || i < lim);
            keep(j1, e1.x, whole || i + 64 < lim);
            keep(j0 + jv, e0.y, whole || i < limb);
            keep(j1 + jv, e1.y, whole || i + 64 < limb);
            continue;
        }
        char *ua = ua0 + (size_t)(a - a0) * step, *ub = ub0 + (size_t)(a - a0) * step;
        if (whole || i < lim) *reinterpret_cast<float *>(ua + b0) = e0.x;
        if (whole || i + 64 < lim) *reinterpret_cast<float *>(ua + b1) = e1.x;
        if (whole || i < limb) *reinterpret_cast<float *>(ub + b0) = e0.y;
        if (whole || i + 64 < limb) *reinterpret_cast<float *>(ub + b1) = e1.y;
    }
}


// ---- a poisoned tile ------------------------------------------------------------------------------------------------
// One inf / nan among a tile's 8192 inputs makes EVERY result of that tile non-finite (each bin of the forward transform is a sum
// over all inputs; no IEEE operation of the transforms turns a non-finite value back into a finite one), where the reference's
// lfilter confines it to the Ntaps outputs that multiply it.  So a wave that finds a non-finite result recomputes the tile's outputs
// by the reference's own sum (careful.hpp: careful_ols_tile) and overwrites what the tile just stored -- after the barrier that ends
// the tile, which orders the two stores to an address whichever threads made them.  The hot path pays two compares per tile for this.
__device__ __forceinline__ OlsCareful ols_careful_args(const OlsArgs &A, int ph, bool cx)
{
    OlsCareful c;
    c.x = A.x; c.y = A.y; c.n = A.n; c.n_hist = A.n_hist; c.n_keep = A.n_keep; c.up_pitch = A.up_pitch;
    c.V = A.V; c.dec = A.dec; c.up = A.up;
    // .up: the interpolation factor and the first phase of this pass, from the byte strides of its store (see OlsArgs)
    const int esz = cx ? 8 : 4;
    c.L = A.up_sb / esz;
    c.p0 = (A.up_pb0 + ph * A.up_pbs) / esz;
    c.cf = A.cf;
    return c;
}

// Persistent: gridDim.x = 2 workgroups per CU, each walks tiles blockIdx.x, +gridDim.x, ...
// Per tile the only vector-memory traffic is [H: 16 loads at tile start, consumed after
// the forward FFT] [next tile's x: 16 loads issued after the H multiply, consumed at the
// next iteration: in flight during the whole inverse FFT] [14 stores].  The
// tile-invariant inter-pass twiddles never touch the VM path (T1 as 15 register-resident
// powers of W_4096^t, T2 as two 4 KiB LDS tables): vmcnt retires in order, so any table
// load issued after a prefetch would force the prefetch to land first.
// DEC: the decimating store (multirate_FIR.dn) is its own instantiation, so that the plain filter carries none of its code
// UP: multirate_FIR.up for phases too long for the polyphase kernels (see OlsArgs::up): the same walk over (tile, phase) pairs, H of the
// pair's phase fetched per pair, outputs stored with stride up.  Neighbouring walk indices are the phases of one input tile: they run on
// one XCD at one time, so the tile is fetched from HBM once and the strided stores of its phases meet in that XCD's L2.
template <bool REAL, bool DEC, bool UP = false, bool XR = false>   // XR: real signal into the complex tile (see load_tile_xr)
__global__ __launch_bounds__(256, 2) void ols_tile_kernel(OlsArgs A)
{
    __shared__ float4 lds[kLdsUnits + 2 * kT2Units];
    __shared__ unsigned long long ols_noted;   // poisoned tiles, by walk step (careful.hpp)
    if (threadIdx.x == 0) ols_noted = 0;       // (the barrier behind the table load lies between this and any note)
    // Decimating store (.dn): the last HS of the thread's 16 H registers are fetched per tile (HS x 4 KiB from L2, requested at the top of the
    // tile, used behind the forward transform).  With all 16 held across tiles hipcc spilled four of them and reloaded them from scratch INSIDE
    // the H multiply: four round trips per tile on the critical path.
    constexpr int HS = (DEC && !UP) ? (REAL ? 4 : 5) : 0;   // (the counts that leave no spill)
    float4 *T2f = lds + kLdsUnits;            // [k2][q]
    float4 *T2t = lds + kLdsUnits + kT2Units; // [qq][k2] (transposed copy for the inverse)
    const int t = threadIdx.x;

    // one-time: LDS twiddle tables and this thread's 15 register twiddles
    {
        const float4 w = A.T2[t];             // t = k2*16 + q
        T2f[t] = w;
        T2t[(t & 15) * 16 + (t >> 4)] = w;
    }
    cf tw[16];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw[k1] = lo(A.T1[k1 * 256 + t]);  // W_8192^(2t k1) = W_4096^(t k1)
    tw[0] = make_float2(1.f, 0.f);

    __syncthreads();
    int it = 0;
    // This thread's 32 bins of H stay in registers for every tile it processes: streaming
    // them per tile cost 64 KiB of L2->CU traffic per tile, a third of everything the CU's
    // vector-memory pipe (~10 B/clk) had to move, and that pipe is what bounds the kernel.
    float4 hh[16];
    // XCD-aware walk: workgroup w runs on XCD w % 8, so give each XCD a contiguous run of tiles per
    // round -- neighbouring tiles share Ntaps-1 input samples, which then hit that XCD's L2 instead of
    // being fetched twice from HBM (the 7.6 % of traffic above the algorithmic bytes)
    int64_t tile = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8
                                                                 : (int64_t)blockIdx.x;
    // (sharded launches walk tile 0 last: walk index w stands for tile w + 1, the last index for tile 0)
    const bool t0_last = A.halo_flag != nullptr;
    // .up: pair w = (input tile w / up, phase w % up); w < 2^31 (checked at launch), so 32-bit divisions
    auto phys = [&](int64_t w) -> int64_t {
        if (UP) return (int64_t)((unsigned)w / (unsigned)A.up);
        return t0_last ? (w + 1 < A.ntiles ? w + 1 : 0) : w;
    };
    auto phase_of = [&](int64_t w) -> int { return (int)((unsigned)w % (unsigned)A.up); };
    if (!UP) load_H(t, A.Hp, hh);
    cf v[32];
    if (tile < A.ntiles) {
        if (t0_last && phys(tile) == 0) wait_halo(A);
        load_any<REAL, XR>(A, phys(tile), t, v);
    }
    // the loop is entered with no load pending on either edge (see the note in front of the stores):
    // waits placed at the loop top for THIS load would otherwise also be paid by every later tile,
    // where the only pending vector-memory operations are the previous tile's stores
    if constexpr (XR) {   // (real signal: only the real parts were loaded -- the zeros must not be pinned into registers here)
#pragma unroll
        for (int i = 0; i < 32; i += 8)
            asm volatile("" ::"v"(v[i].x), "v"(v[i + 1].x), "v"(v[i + 2].x), "v"(v[i + 3].x), "v"(v[i + 4].x), "v"(v[i + 5].x), "v"(v[i + 6].x), "v"(v[i + 7].x));
    } else {
#pragma unroll
    for (int i = 0; i < 32; i += 8)
        asm volatile("" ::"v"(v[i].x), "v"(v[i].y), "v"(v[i + 1].x), "v"(v[i + 1].y), "v"(v[i + 2].x), "v"(v[i + 2].y),
                     "v"(v[i + 3].x), "v"(v[i + 3].y), "v"(v[i + 4].x), "v"(v[i + 4].y), "v"(v[i + 5].x), "v"(v[i + 5].y),
                     "v"(v[i + 6].x), "v"(v[i + 6].y), "v"(v[i + 7].x), "v"(v[i + 7].y));
    }
    for (; tile < A.ntiles; tile += gridDim.x, ++it) {
        if (HS > 0) {
            int tt = t;   // (opaque: the requests stay inside the tile loop)
            asm volatile("" : "+v"(tt));
#pragma unroll
            for (int j = 16 - HS; j < 16; ++j) hh[j] = vld(reinterpret_cast<const volatile float4 *>(A.Hp) + (unsigned)(j * 256 + tt));
        }
        if (UP) {   // this pair's phase (streamed per pair: 64 KiB from L2, requested here, used behind the forward transform.  One phase per
                    // workgroup with H held in registers like .filter -- a grid that is a multiple of up -- was built and measured:
                    // 0.66 vs 0.52 ms at L = 12, 0.39 vs 0.35 at L = 4, equal at L = 2; never faster, removed)
            int tt = t;
            asm volatile("" : "+v"(tt));
            const volatile float4 *hp = reinterpret_cast<const volatile float4 *>(A.Hp) + (size_t)phase_of(tile) * 4096;
#pragma unroll
            for (int j = 0; j < 16; ++j) hh[j] = vld(hp + (unsigned)(j * 256 + tt));
        }
        if (REAL) fwd_pass1_real(t, v, tw, lds); else fwd_pass1(t, v, tw, lds);
        __syncthreads();
        cf Z[32];
        fwd_pass23(t, T2f, lds, Z);
        mul_H(hh, Z);
        // x of the next tile: requested now, consumed at the top of the next iteration -- in flight during the
        // whole inverse FFT (requested one phase earlier, right behind pass 1: 61 spilled VGPRs in the float32 variant and
        // 0.240 vs 0.236 ms for complex64: not kept).  The wave runs at raised priority while it issues memory
        // instructions (here and at the stores): -1.1 % (0.2314 vs 0.2339 ms, alternating runs)
        const int64_t next = tile + gridDim.x;
        cf nx[32];
        __builtin_amdgcn_s_setprio(3);
        if (next < A.ntiles) {
            if (t0_last && phys(next) == 0) wait_halo(A);  // (uniform: the whole workgroup owns that tile)
            load_any<REAL, XR>(A, phys(next), t, nx);
        }
        __builtin_amdgcn_s_setprio(0);
        inv_pass32(t, T2t, lds, Z);
        __syncthreads();
        inv_pass1(t, tw, lds, v);
        // Have hipcc wait for the prefetch HERE, while only loads are outstanding (they have had the
        // whole inverse FFT to land).  Otherwise the wait sits at the top of the next tile, behind the
        // stores below, and vmcnt -- which retires in order -- makes every tile start with a full
        // round trip of its predecessor's stores.
        if constexpr (XR) {
#pragma unroll
            for (int i = 0; i < 32; i += 8)
                asm volatile("" ::"v"(nx[i].x), "v"(nx[i + 1].x), "v"(nx[i + 2].x), "v"(nx[i + 3].x), "v"(nx[i + 4].x), "v"(nx[i + 5].x), "v"(nx[i + 6].x), "v"(nx[i + 7].x)
                             : "memory");
        } else {
#pragma unroll
        for (int i = 0; i < 32; i += 8)
            asm volatile("" ::"v"(nx[i].x), "v"(nx[i].y), "v"(nx[i + 1].x), "v"(nx[i + 1].y), "v"(nx[i + 2].x), "v"(nx[i + 2].y),
                         "v"(nx[i + 3].x), "v"(nx[i + 3].y), "v"(nx[i + 4].x), "v"(nx[i + 4].y), "v"(nx[i + 5].x), "v"(nx[i + 5].y),
                         "v"(nx[i + 6].x), "v"(nx[i + 6].y), "v"(nx[i + 7].x), "v"(nx[i + 7].y)
                         : "memory");
        }
        __builtin_amdgcn_s_setprio(3);
        if (UP) {
            if (!DEC && A.up_pitch) {   // phases as rows: the plain filter's full-width stores into row phase_of(tile)
                OlsArgs B = A;
                B.y = REAL ? reinterpret_cast<cf *>(reinterpret_cast<float *>(A.y) + (int64_t)phase_of(tile) * A.up_pitch) : A.y + (int64_t)phase_of(tile) * A.up_pitch;
                store_any<REAL, false>(B, phys(tile), t, v, lds);
            } else if (REAL) {
                store_tile_real_up<DEC>(A, phys(tile), phase_of(tile), t, v);
            } else {
                store_tile_up<DEC>(A, phys(tile), phase_of(tile), t, v);
            }
        } else {
            store_any<REAL, DEC>(A, phys(tile), t, v, lds);
        }
        __builtin_amdgcn_s_setprio(0);
        // (all results of a tile are non-finite, or none; noted by walk step and recomputed behind the loop: careful.hpp)
        if (__builtin_expect(__any(not_finite(v[31].x) | not_finite(v[31].y)), 0)) careful_note(&ols_noted, it);
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = nx[i];
        // Every wave must be done reading the image before the next tile overwrites it -- unless nobody else ever touches what a thread overwrites:
        // inv_pass1 of thread (b, q) reads lds_unit(k1, b, q), k1 = 0 .. 15, and fwd_pass1 of the same thread writes exactly those sixteen units (the
        // inverse is the mirror image of the forward transform).  Between barrier 2 of this tile and barrier 1 of the next a thread therefore meets
        // only its own units -- read, then overwritten, in program order -- and the third barrier of the plain complex tile is not needed (round 6).
        // The other forms keep it: the two-real-tiles pass 1 writes another set of units, the decimating store gathers in the idle image, and the
        // interpolating stores of complex tiles share this loop with forms that do.
#ifdef SK_OLS_KEEP_B3   // (A/B builds)
        __syncthreads();
#else
        if constexpr (REAL || DEC || UP || XR) __syncthreads();
#endif
    }
    const unsigned long long noted = careful_noted(&ols_noted);
    if (__builtin_expect(noted != 0, 0)) {
        // (the walk again, as the loop made it)
        int64_t w = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
        for (int64_t k = 0; w < A.ntiles; w += gridDim.x, ++k)
            if (careful_step_noted(noted, k))
                careful_ols_tile<float, REAL, DEC, UP, XR>(ols_careful_args(A, UP ? phase_of(w) : 0, !(REAL || XR)), phys(w), UP ? phase_of(w) : 0, t);
    }
}


// ---- multirate_FIR.dn, even M: the decimating INVERSE transform ----------------------------------------------------------------------
// The decimating store above computes every full-rate output and keeps one in M.  Here the spectrum is folded M-fold between the
// registers of a thread and only the kept outputs are transformed back (ols_core.hpp: inv_pass32_fold) -- 1 forward + ~1/4 inverse
// transform per tile instead of 2; loads, prefetch, H product and the walk are the plain filter's.  Thread (b, q) with q a multiple
// of MF / 2 ends up with y[512 a + 32 b + 2 q], a = 0 .. 15: element (512 / MF) a + (32 / MF) b + 2 q / MF of the tile's MF-fold decimated
// run, so for every a the workgroup's active lanes hold 512 / MF consecutive elements.  MF = the largest of 16, 8, 4, 2 that divides M;
// what is left of M (A.dec / MF = 3 for the reference's default M = 12) is taken at the store: every (M / MF)-th element of the run
// leaves.  REAL: two real tiles ride in one complex tile.
template <bool REAL, int MF>
__global__ __launch_bounds__(256, 2) void ols_fold_kernel(OlsArgs A)
{
    __shared__ float4 lds[kLdsUnits + 2 * kT2Units];
    __shared__ unsigned long long ols_noted;   // poisoned tiles, by walk step (careful.hpp)
    if (threadIdx.x == 0) ols_noted = 0;
    float4 *T2f = lds + kLdsUnits, *T2t = lds + kLdsUnits + kT2Units;
    const int t = threadIdx.x;
    {
        const float4 w = A.T2[t];
        T2f[t] = w;
        T2t[(t & 15) * 16 + (t >> 4)] = w;
    }
    cf tw[16];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw[k1] = lo(A.T1[k1 * 256 + t]);
    tw[0] = make_float2(1.f, 0.f);
    __syncthreads();
    float4 hh[16];
    auto first_tile = [&]() -> int64_t { return (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x; };
    int64_t tile = first_tile();
    load_H(t, A.Hp, hh);
    cf v[32];
    if (tile < A.ntiles) load_any<REAL, false>(A, tile, t, v);
#pragma unroll
    for (int i = 0; i < 32; i += 8)
        asm volatile("" ::"v"(v[i].x), "v"(v[i].y), "v"(v[i + 1].x), "v"(v[i + 1].y), "v"(v[i + 2].x), "v"(v[i + 2].y),
                     "v"(v[i + 3].x), "v"(v[i + 3].y), "v"(v[i + 4].x), "v"(v[i + 4].y), "v"(v[i + 5].x), "v"(v[i + 5].y),
                     "v"(v[i + 6].x), "v"(v[i + 6].y), "v"(v[i + 7].x), "v"(v[i + 7].y));
    constexpr int LS = MF / 2;
    const int b = t >> 4, q = t & 15;
    const bool active = q % LS == 0;
    // this lane's first kept output inside a tile (a = 0), and how many kept outputs a step of a spans
    const int j0 = (32 / MF) * b + q / LS;
    constexpr int JA = 512 / MF;
    for (int64_t step = 0; tile < A.ntiles; tile += gridDim.x, ++step) {
        if (REAL) fwd_pass1_real(t, v, tw, lds); else fwd_pass1(t, v, tw, lds);
        __syncthreads();
        cf Z[32];
        fwd_pass23(t, T2f, lds, Z);
        mul_H(hh, Z);
        const int64_t next = tile + gridDim.x;
        cf nx[32];
        __builtin_amdgcn_s_setprio(3);
        if (next < A.ntiles) load_any<REAL, false>(A, next, t, nx);
        __builtin_amdgcn_s_setprio(0);
        inv_pass32_fold<MF>(t, T2t, lds, Z);
        __syncthreads();
        cf o[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) o[a] = make_float2(0.f, 0.f);
        inv_pass1_fold<MF>(t, tw, lds, o);
#pragma unroll
        for (int i = 0; i < 32; i += 8)
            asm volatile("" ::"v"(nx[i].x), "v"(nx[i].y), "v"(nx[i + 1].x), "v"(nx[i + 1].y), "v"(nx[i + 2].x), "v"(nx[i + 2].y),
                         "v"(nx[i + 3].x), "v"(nx[i + 3].y), "v"(nx[i + 4].x), "v"(nx[i + 4].y), "v"(nx[i + 5].x), "v"(nx[i + 5].y),
                         "v"(nx[i + 6].x), "v"(nx[i + 6].y), "v"(nx[i + 7].x), "v"(nx[i + 7].y)
                         : "memory");
        // ---- the kept outputs of the tile (pair of tiles): from block a0 on, V / MF elements of the decimated run per tile, every Mr-th kept ----
        __builtin_amdgcn_s_setprio(3);
        if (active) {
            int a0 = A.a0;
            asm volatile("" : "+s"(a0));
            const int Mr = A.dec / MF;
            const int64_t n_out = A.n_keep / A.dec;
            const int VD = A.V / MF;                                       // elements of the decimated run per tile
            // element e of the run is output e / Mr where Mr divides e: the tile's (pair's) first element = Mr q0 + r0
            const int64_t e0 = (REAL ? 2 * tile : tile) * (int64_t)VD;
            const int64_t q0 = Mr > 1 ? e0 / Mr : e0;
            const unsigned r0 = Mr > 1 ? (unsigned)(e0 - q0 * Mr) : 0u;
            auto put = [&](unsigned el, auto val, auto *yp) __attribute__((always_inline)) {   // el: element relative to e0
                if (Mr > 1) {
                    const unsigned g = r0 + el;
                    const unsigned k = (unsigned)(((unsigned long long)g * A.dec_magic) >> 32);   // g / Mr (dec_magic = ceil(2^32 / Mr) here)
                    if (k * (unsigned)Mr == g && q0 + k < n_out) __builtin_nontemporal_store(val, yp + (q0 + k));
                } else if (q0 + el < n_out) {
                    __builtin_nontemporal_store(val, yp + (q0 + el));
                }
            };
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                if (a < a0) continue;
                const unsigned el = (unsigned)(j0 + JA * (a - a0));
                if (REAL) {
                    put(el, o[a].x, reinterpret_cast<float *>(A.y));
                    put(el + (unsigned)VD, o[a].y, reinterpret_cast<float *>(A.y));
                } else {
                    put(el, v2f_t{o[a].x, o[a].y}, reinterpret_cast<v2f_t *>(A.y));
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // (all results of a tile are non-finite, or none; noted by walk step and recomputed behind the loop: careful.hpp)
        if (__builtin_expect(__any(active && (not_finite(o[15].x) | not_finite(o[15].y))), 0)) careful_note(&ols_noted, step);
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = nx[i];
        // (complex tiles: the units thread (b, q) reads in the folded inverse pass 1 are among the sixteen its next forward pass 1 overwrites, and nobody
        // else's -- no third barrier, as in ols_tile_kernel; the two-real-tiles pass 1 writes another set and keeps it)
#ifdef SK_OLS_KEEP_B3
        __syncthreads();
#else
        if constexpr (REAL) __syncthreads();
#endif
    }
    const unsigned long long noted = careful_noted(&ols_noted);
    if (__builtin_expect(noted != 0, 0)) {
        int64_t w = first_tile();
        for (int64_t k = 0; w < A.ntiles; w += gridDim.x, ++k)
            if (careful_step_noted(noted, k)) careful_ols_tile<float, REAL, true, false, false>(ols_careful_args(A, 0, !REAL), w, 0, t);
    }
}


// ---- multirate_FIR.up, even L: the forward transform of the zero-stuffed tile from its non-zero columns alone ---------------------------------
// (ols_core.hpp: fwd_pass1_rep / fwd_pass23_rep.)  The tile is a tile of the OUTPUT: loads are a quarter (L = 4) of the plain filter's, the forward
// transform about a quarter, H product, inverse transform and the full-width 16-byte stores the plain filter's own.  LF = the largest of 16, 8,
// 4, 2 that divides L; with L = LF rep_lr (12 = 4 x 3) the LF-fold decimated grid is itself zero-stuffed: only every rep_lr-th of its samples is an
// input.  Thread (b, q) with q a multiple of LF / 2 loads grid elements (512 / LF) a + (32 / LF) b + 2 q / LF of the tile, a = 0 .. 15.
template <bool REAL, int LF>
__device__ __forceinline__ void load_rep(const OlsArgs &A, int64_t tile, int t, cf *in)
{
    constexpr int LS = LF / 2, JA = 512 / LF;
    const int b = t >> 4, q = t & 15;
#pragma unroll
    for (int a = 0; a < 16; ++a) in[a] = make_float2(0.f, 0.f);
    if (q % LS != 0) return;
    const int j0 = (32 / LF) * b + q / LS;
    const int Lr = A.rep_lr;
    const int64_t in0 = (REAL ? 2 * tile : tile) * (int64_t)A.V - A.ov;    // up-rate index of the tile's first sample (a multiple of 512)
    const int64_t e0 = in0 / LF;                                            // the same on the LF-fold decimated grid (exact)
    // grid element e is input sample e / Lr where Lr divides e: e0 = Lr q0 + r0, 0 <= r0 < Lr (floor division: e0 is negative in the first tile)
    int64_t q0 = e0, r0 = 0;
    if (Lr > 1) {
        q0 = e0 / Lr;
        r0 = e0 - q0 * Lr;
        if (r0 < 0) { r0 += Lr; q0 -= 1; }
    }
    const int VD = A.V / LF;                                                // grid elements between the pair's two tiles (REAL)
    auto fetch = [&](unsigned d, int64_t &idx) -> bool {                    // grid element e0 + d -> input index; false: a stuffed zero
        if (Lr == 1) { idx = q0 + d; return true; }
        const unsigned g = (unsigned)r0 + d;
        const unsigned k = (unsigned)(((unsigned long long)g * A.rep_magic) >> 32);
        idx = q0 + k;
        return k * (unsigned)Lr == g;
    };
    const bool interior = Lr == 1 && A.aligned && e0 >= -A.rep_hist && e0 + kN / LF + (REAL ? VD : 0) <= A.n_in;
    int tt = j0;   // (opaque copy: the addresses are rebuilt per tile instead of living in registers across the tile loop)
    asm volatile("" : "+v"(tt));
    if (interior) {
        if (REAL) {
            const float *xa = reinterpret_cast<const float *>(A.x) + e0, *xb = xa + VD;
#pragma unroll
            for (int a = 0; a < 16; ++a) in[a] = make_float2(__builtin_nontemporal_load(xa + (unsigned)(JA * a + tt)), __builtin_nontemporal_load(xb + (unsigned)(JA * a + tt)));
        } else {
            const v2f_t *xc = reinterpret_cast<const v2f_t *>(A.x) + e0;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                const v2f_t r = __builtin_nontemporal_load(xc + (unsigned)(JA * a + tt));
                in[a] = make_float2(r.x, r.y);
            }
        }
        return;
    }
#pragma unroll   // (unrolled: a run-time index into `in` would move the whole array to scratch memory)
    for (int a = 0; a < 16; ++a) {
        int64_t ia, ib;
        if (REAL) {
            const float *xr = reinterpret_cast<const float *>(A.x);
            float va = 0.f, vb = 0.f;
            if (fetch((unsigned)(JA * a + tt), ia) && ia >= -A.rep_hist && ia < A.n_in) va = xr[ia];
            if (fetch((unsigned)(JA * a + tt + VD), ib) && ib >= -A.rep_hist && ib < A.n_in) vb = xr[ib];
            in[a] = make_float2(va, vb);
        } else if (fetch((unsigned)(JA * a + tt), ia) && ia >= -A.rep_hist && ia < A.n_in) {
            in[a] = A.x[ia];
        }
    }
}

template <bool REAL, int LF>
__global__ __launch_bounds__(256, 2) void ols_rep_kernel(OlsArgs A)
{
    __shared__ float4 lds[kLdsUnits + 2 * kT2Units];
    __shared__ unsigned long long ols_noted;   // poisoned tiles, by walk step (careful.hpp)
    if (threadIdx.x == 0) ols_noted = 0;
    float4 *T2f = lds + kLdsUnits, *T2t = lds + kLdsUnits + kT2Units;
    const int t = threadIdx.x;
    {
        const float4 w = A.T2[t];
        T2f[t] = w;
        T2t[(t & 15) * 16 + (t >> 4)] = w;
    }
    cf tw[16];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw[k1] = lo(A.T1[k1 * 256 + t]);
    tw[0] = make_float2(1.f, 0.f);
    __syncthreads();
    float4 hh[16];
    auto first_tile = [&]() -> int64_t { return (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x; };
    int64_t tile = first_tile();
    load_H(t, A.Hp, hh);
    cf in[16];
    if (tile < A.ntiles) load_rep<REAL, LF>(A, tile, t, in);
#pragma unroll
    for (int i = 0; i < 16; i += 8)
        asm volatile("" ::"v"(in[i].x), "v"(in[i].y), "v"(in[i + 1].x), "v"(in[i + 1].y), "v"(in[i + 2].x), "v"(in[i + 2].y),
                     "v"(in[i + 3].x), "v"(in[i + 3].y), "v"(in[i + 4].x), "v"(in[i + 4].y), "v"(in[i + 5].x), "v"(in[i + 5].y),
                     "v"(in[i + 6].x), "v"(in[i + 6].y), "v"(in[i + 7].x), "v"(in[i + 7].y));
    for (int64_t step = 0; tile < A.ntiles; tile += gridDim.x, ++step) {
        fwd_pass1_rep<LF>(t, in, tw, lds);
        __syncthreads();
        cf Z[32];
        fwd_pass23_rep<LF>(t, T2f, lds, Z);
        mul_H(hh, Z);
        const int64_t next = tile + gridDim.x;
        cf nx[16];
        __builtin_amdgcn_s_setprio(3);
        if (next < A.ntiles) load_rep<REAL, LF>(A, next, t, nx);
        __builtin_amdgcn_s_setprio(0);
        inv_pass32(t, T2t, lds, Z);
        __syncthreads();
        cf v[32];
        inv_pass1(t, tw, lds, v);
#pragma unroll
        for (int i = 0; i < 16; i += 8)
            asm volatile("" ::"v"(nx[i].x), "v"(nx[i].y), "v"(nx[i + 1].x), "v"(nx[i + 1].y), "v"(nx[i + 2].x), "v"(nx[i + 2].y),
                         "v"(nx[i + 3].x), "v"(nx[i + 3].y), "v"(nx[i + 4].x), "v"(nx[i + 4].y), "v"(nx[i + 5].x), "v"(nx[i + 5].y),
                         "v"(nx[i + 6].x), "v"(nx[i + 6].y), "v"(nx[i + 7].x), "v"(nx[i + 7].y)
                         : "memory");
        __builtin_amdgcn_s_setprio(3);
        store_any<REAL, false>(A, tile, t, v, lds);
        __builtin_amdgcn_s_setprio(0);
        if (__builtin_expect(__any(not_finite(v[31].x) | not_finite(v[31].y)), 0)) careful_note(&ols_noted, step);
#pragma unroll
        for (int i = 0; i < 16; ++i) in[i] = nx[i];
        // (the barrier is not NEEDED for complex tiles -- the pruned forward pass 1 of thread (b, q) writes into the units its inverse pass 1 has just read,
        // and nobody else's -- but without it this kernel measured 1 % slower, same box, alternating: 0.1712 -> 0.1732 ms for L = 4; kept)
        __syncthreads();
    }
    const unsigned long long noted = careful_noted(&ols_noted);
    if (__builtin_expect(noted != 0, 0)) {
        int64_t w = first_tile();
        for (int64_t k = 0; w < A.ntiles; w += gridDim.x, ++k)
            if (careful_step_noted(noted, k))
                careful_fir_range<float, !REAL>(A.x, A.y, A.rep_hist, A.n, (REAL ? 2 * w : w) * (int64_t)A.V, (REAL ? 2 : 1) * (int64_t)A.V, A.rep_L, 1, A.cf, t);
    }
}

bool fir_ols_supported(const FirHandle *h)
{
    // complex64 signal; overlap must leave at least half the tile as useful output
    if (h->ntaps < 2 || h->ntaps - 1 > 4096) return false;
    // complex64 with any taps; float32 with real taps (two real tiles per complex tile)
    return h->dtype == SKDSP_C64 || (h->dtype == SKDSP_F32 && !h->taps_complex);
}

// Tables of one plan: `up` phase filters (phase p: taps gain * b[p + up t], t < T) as `up` consecutive Hp tables; up = 1 is the filter itself.
// kind 0: all `up` phases; 1: pairs of phases (real taps: table k holds phase 2k + i phase 2k+1, see load_tile_xr; an odd up leaves its last
// phase out); 2: the last phase alone (what kind 1 leaves out)
static int build_plan(const FirHandle *h, int up, OlsPlan **out, int kind = 0)
{
    if (kind == 1) {
        const int T = (h->ntaps + up - 1) / up;
        OlsPlan *p = new OlsPlan();
        p->ntaps = T;
        p->ov = ((T - 1 + 511) / 512) * 512;
        if (p->ov == 0) p->ov = 512;
        p->V = kN - p->ov;
        std::vector<float4> T1, T2, Hp, Hall;
        make_T1(T1);
        make_T2(T2);
        std::vector<double> ph((size_t)T * 2);
        for (int k = 0; k < up / 2; ++k) {
            std::fill(ph.begin(), ph.end(), 0.0);
            for (int t = 0; t < T; ++t)
                for (int c = 0; c < 2; ++c) {
                    const int j = 2 * k + c + up * t;
                    if (j < h->ntaps) ph[(size_t)2 * t + c] = (double)up * h->taps_host[j];
                }
            make_Hp(ph.data(), T, 2, Hp);
            Hall.insert(Hall.end(), Hp.begin(), Hp.end());
        }
        hipError_t e;
        if ((e = hipMalloc((void **)&p->T1, T1.size() * sizeof(float4))) != hipSuccess ||
            (e = hipMalloc((void **)&p->T2, T2.size() * sizeof(float4))) != hipSuccess ||
            (e = hipMalloc((void **)&p->Hp, Hall.size() * sizeof(float4))) != hipSuccess ||
            (e = hipMemcpy(p->T1, T1.data(), T1.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(p->T2, T2.data(), T2.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(p->Hp, Hall.data(), Hall.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess) {
            fir_ols_free(p);
            return hip_fail(e, "ols tables (paired phases)", __FILE__, __LINE__);
        }
        *out = p;
        return SKDSP_OK;
    }
    const int comp = h->taps_complex ? 2 : 1;
    const int T = kind == 3 ? h->ntaps : (h->ntaps + up - 1) / up;   // (kind 3: the WHOLE filter at the high rate, with the gain L: ols_rep_kernel)
    const int q_first = kind == 2 ? up - 1 : 0;
    OlsPlan *p = new OlsPlan();
    p->ntaps = T;
    p->ov = ((T - 1 + 511) / 512) * 512;
    if (p->ov == 0) p->ov = 512;
    p->V = kN - p->ov;
    std::vector<float4> T1, T2, Hp, Hall;
    make_T1(T1);
    make_T2(T2);
    if (up == 1) {
        make_Hp(h->taps_host.data(), h->ntaps, comp, Hall);
    } else if (kind == 3) {
        std::vector<double> scaled(h->taps_host);
        for (double &v : scaled) v *= (double)up;   // (the gain L of multirate_FIR.up)
        make_Hp(scaled.data(), h->ntaps, comp, Hall);
    } else {
        std::vector<double> ph((size_t)T * comp);
        for (int q = q_first; q < up; ++q) {
            std::fill(ph.begin(), ph.end(), 0.0);
            for (int t = 0; t < T; ++t) {
                const int k = q + up * t;
                if (k >= h->ntaps) break;
                for (int c = 0; c < comp; ++c) ph[(size_t)t * comp + c] = (double)up * h->taps_host[(size_t)k * comp + c];  // (the gain L of multirate_FIR.up)
            }
            make_Hp(ph.data(), T, comp, Hp);
            Hall.insert(Hall.end(), Hp.begin(), Hp.end());
        }
    }
    hipError_t e;
    if ((e = hipMalloc((void **)&p->T1, T1.size() * sizeof(float4))) != hipSuccess ||
        (e = hipMalloc((void **)&p->T2, T2.size() * sizeof(float4))) != hipSuccess ||
        (e = hipMalloc((void **)&p->Hp, Hall.size() * sizeof(float4))) != hipSuccess) {
        fir_ols_free(p);
        return hip_fail(e, "hipMalloc(ols tables)", __FILE__, __LINE__);
    }
    if ((e = hipMemcpy(p->T1, T1.data(), T1.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->T2, T2.data(), T2.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->Hp, Hall.data(), Hall.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess) {
        fir_ols_free(p);
        return hip_fail(e, "hipMemcpy(ols tables)", __FILE__, __LINE__);
    }
    *out = p;
    return SKDSP_OK;
}

static int ensure_plan(FirHandle *h)
{
    if (h->ols) return SKDSP_OK;
    return build_plan(h, 1, &h->ols);
}

void fir_ols_free(OlsPlan *p)
{
    if (!p) return;
    if (p->T1) (void)hipFree(p->T1);
    if (p->T2) (void)hipFree(p->T2);
    if (p->Hp) (void)hipFree(p->Hp);
    delete p;
}

int fir_ols_tile_outputs(FirHandle *h, int *V)
{
    int rc = ensure_plan(h);
    if (rc) return rc;
    *V = h->ols->V;
    return SKDSP_OK;
}

__global__ void ols_halo_flag_kernel(unsigned *flag, unsigned seq)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int fir_ols_publish_halo(unsigned *flag, unsigned seq, hipStream_t s)
{
    hipLaunchKernelGGL(ols_halo_flag_kernel, dim3(1), dim3(1), 0, s, flag, seq);
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

int fir_ols_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, void *y, hipStream_t s, int dec, int reserve_wgs,
                   const unsigned *halo_flag, unsigned halo_seq, unsigned *halo_err, int halo_spins)
{
    note_path("fir_ols");
    if (n <= 0) return SKDSP_OK;
    if (dec > 1) n = (n / dec) * dec;  // the dropped tail is never computed
    if (n <= 0) return SKDSP_OK;
    SK_CHECK(fir_ols_supported(h), SKDSP_ERR_UNSUPPORTED, "fir_ols: needs complex64 (or float32 with real taps) and 2..4097 taps");
    // (the decimating stores divide tile-local indices below M + 16384 by M through a multiply-high by ceil(2^32 / M): exact while (M + 16384) M < 2^32)
    SK_CHECK(dec <= 32768, SKDSP_ERR_UNSUPPORTED, "fir_ols: decimation by %d (the decimating store takes M <= 32768)", dec);
    int rc = ensure_plan(h);
    if (rc) return rc;
    OlsPlan *p = h->ols;
    OlsArgs A;
    A.x = (const cf *)x;
    A.y = (cf *)y;
    A.n = n;
    A.n_hist = n_hist;
    A.T1 = p->T1; A.T2 = p->T2; A.Hp = p->Hp;
    A.ov = p->ov; A.V = p->V; A.a0 = p->ov / 512; A.keep = opt().ols_keep_overlap ? A.a0 : 0;
    const bool real = h->dtype == SKDSP_F32;
    // element alignment is all the vector accesses need: tile starts are odd multiples of the
    // element size anyway (V = 8192 - (Ntaps-1) is odd for even tap counts)
    A.aligned = ((((uintptr_t)x) | ((uintptr_t)y)) & (real ? 3 : 7)) == 0;
    int64_t ntiles = (n + p->V - 1) / p->V;
    if (real) ntiles = (ntiles + 1) / 2;  // pairs of real tiles
    SK_CHECK(ntiles < (int64_t)1 << 31, SKDSP_ERR_BADARG, "fir_ols: too many tiles");
    A.ntiles = ntiles;
    A.dec = dec > 1 ? dec : 1;
    A.dec_magic = A.dec > 1 ? (unsigned)((((unsigned long long)1 << 32) + A.dec - 1) / A.dec) : 0u;
    A.n_keep = n;
    A.up = 1; A.up_pitch = 0; A.up_sb = A.up_pbs = A.up_pb0 = 0;
    A.halo_flag = halo_flag; A.halo_seq = halo_seq; A.halo_err = halo_err; A.halo_spins = halo_spins > 0 ? halo_spins : (1 << 21);
    A.rep_L = 0; A.rep_lr = 1; A.rep_magic = 0; A.n_in = 0; A.rep_hist = 0;
    if ((rc = fir_careful(h, &A.cf))) return rc;
    SK_CHECK(!(halo_flag && real), SKDSP_ERR_UNSUPPORTED, "fir_ols: halo flag wait is for complex64 shards");
    int64_t grid = 2 * (int64_t)ctx().num_cus;  // 2 resident workgroups per CU (76 KiB LDS each)
    // 8 slots stay free: room for a concurrent kernel (the RCCL send/recv of a halo, another stream of the caller) at
    // no measurable cost (0.2375 vs 0.2375 ms at 2^26, alternating runs on one box)
    if (reserve_wgs < 0) reserve_wgs = opt().ols_reserve;
    if (reserve_wgs > 0 && grid >= 4 * (int64_t)reserve_wgs) grid -= reserve_wgs;
    if (grid > ntiles) grid = ntiles;
    const int mf = A.dec % 16 == 0 ? 16 : (A.dec % 8 == 0 ? 8 : (A.dec % 4 == 0 ? 4 : (A.dec % 2 == 0 ? 2 : 1)));
    const bool fold = opt().fir_dn_fold && !halo_flag && mf > 1;
    if (fold) {   // the decimating inverse transform (ols_fold_kernel): M = mf x (what the store takes)
        const int mr = A.dec / mf;
        A.dec_magic = mr > 1 ? (unsigned)((((unsigned long long)1 << 32) + mr - 1) / mr) : 0u;   // (elements of the decimated run: below 2^14 + mr, exact for mr <= 4096)
#define SK_FOLD(MF)                                                                                                  \
    if (real) hipLaunchKernelGGL((ols_fold_kernel<true, MF>), dim3((unsigned)grid), dim3(256), 0, s, A);             \
    else hipLaunchKernelGGL((ols_fold_kernel<false, MF>), dim3((unsigned)grid), dim3(256), 0, s, A)
        switch (mf) {
        case 2: SK_FOLD(2); break;
        case 4: SK_FOLD(4); break;
        case 8: SK_FOLD(8); break;
        default: SK_FOLD(16); break;
        }
#undef SK_FOLD
    } else if (A.dec > 1) {
        if (real) hipLaunchKernelGGL((ols_tile_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((ols_tile_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
    } else {
        if (real) hipLaunchKernelGGL((ols_tile_kernel<true, false>), dim3((unsigned)grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((ols_tile_kernel<false, false>), dim3((unsigned)grid), dim3(256), 0, s, A);
    }
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

// ---- multirate_FIR.up with long phases ---------------------------------------------------------------------------
// y[i L + p] = L sum_t b[p + L t] x[i - t]: L filters of ceil(Ntaps / L) taps over the SAME input, outputs interleaved.  The polyphase
// kernels (fir_direct / fir_bx) spend Ntaps / L multiply-adds per output; from ~100 taps per phase on, the overlap-save walk over (tile,
// phase) pairs is cheaper: every pair costs what one tile of .filter costs, whatever the phase length.
bool fir_ols_up_supported(const FirHandle *h, int L)
{
    if (L < 2 || L > 256) return false;   // (the every-M-th store: L <= 64, checked at launch)
    const int T = (h->ntaps + L - 1) / L;
    if (T < 2 || T - 1 > 4096) return false;
    return h->dtype == SKDSP_C64 || (h->dtype == SKDSP_F32 && !h->taps_complex);
}

// float32 signals, no decimation: the phases run in pairs through the complex tile (load_tile_xr).  Even L: an 8-byte aligned destination
// (L / 2 rows of pairs in the rows form).  Odd L (7 .. 13, strided form only): (L - 1) / 2 pairs as 8-byte elements at 4-byte aligned
// addresses, then the last phase on its own (two real tiles per pass) -- L passes per two tiles either way.  Measured against one phase
// per pass (2^26 outputs, 256 taps per phase): L = 7 0.237 -> 0.216 ms, 9 0.248 (rows) -> 0.209, 11 0.255 -> 0.229, 13 0.254 -> 0.245;
// L = 3, 5 lose (0.168 -> 0.186, 0.200 -> 0.208: the misaligned 8-byte stores and the second launch), 15 loses to the rows form.
bool fir_ols_up_pairs(const FirHandle *h, int L, int dec, const void *y)
{
    if (!opt().fir_up_pair || h->dtype != SKDSP_F32 || h->taps_complex || dec > 1) return false;
    if (L % 2 == 0) return ((uintptr_t)y & 7) == 0;
    return L >= 7 && L <= 13 && ((uintptr_t)y & 3) == 0;
}

static int up_plan(FirHandle *h, int L, int kind, OlsPlan **out)
{
    const int key = kind == 1 ? -L : (kind == 2 ? 1000 + L : (kind == 3 ? 100000 + L : L));
    for (auto &u : h->ols_up)
        if (u.L == key) { *out = u.plan; return SKDSP_OK; }
    int rc = build_plan(h, L, out, kind);
    if (rc) return rc;
    h->ols_up.push_back(FirHandle::OlsUp{key, *out});
    return SKDSP_OK;
}

// one launch of the walk over `cnt` phases (kind as in build_plan)
static int up_walk(FirHandle *h, int kind, const void *x, int64_t n, int64_t n_hist, int L, void *y, hipStream_t s, int dec, int64_t rows_pitch)
{
    OlsPlan *p = nullptr;
    int rc = up_plan(h, L, kind, &p);
    if (rc) return rc;
    const bool xr = kind == 1;                              // real signal into the complex tile
    const bool real = h->dtype == SKDSP_F32 && !xr;         // two real tiles per complex tile
    const int cnt = kind == 1 ? L / 2 : (kind == 2 ? 1 : L);
    OlsArgs A;
    A.x = (const cf *)x;
    A.y = (cf *)y;
    A.n = n;
    A.n_hist = n_hist;
    A.T1 = p->T1; A.T2 = p->T2; A.Hp = p->Hp;
    A.ov = p->ov; A.V = p->V; A.a0 = p->ov / 512; A.keep = opt().ols_keep_overlap ? A.a0 : 0;
    const int esz = h->dtype == SKDSP_F32 ? 4 : 8;
    A.aligned = xr ? (((uintptr_t)x & 3) == 0 && ((uintptr_t)y & (L % 2 ? 3 : 7)) == 0) : ((((uintptr_t)x) | ((uintptr_t)y)) & (real ? 3 : 7)) == 0;
    int64_t ntiles = (n + p->V - 1) / p->V;
    if (real) ntiles = (ntiles + 1) / 2;
    ntiles *= cnt;
    SK_CHECK(ntiles < (int64_t)1 << 31, SKDSP_ERR_BADARG, "fir_ols_up: too many tiles");
    A.ntiles = ntiles;
    A.dec = dec;
    A.dec_magic = dec > 1 ? (unsigned)((((unsigned long long)1 << 32) + dec - 1) / dec) : 0u;
    A.n_keep = dec > 1 ? (n * L) / dec : n;   // (DEC: the number of outputs)
    A.up = cnt;   // (DEC launches cover all L phases: there it is also the interpolation factor of the up-rate index)
    A.up_pitch = dec > 1 ? 0 : rows_pitch;
    A.up_sb = L * esz;
    A.up_pbs = xr ? 8 : esz;
    A.up_pb0 = kind == 2 ? (L - 1) * esz : 0;
    A.halo_flag = nullptr; A.halo_seq = 0; A.halo_err = nullptr; A.halo_spins = 0;
    A.rep_L = 0; A.rep_lr = 1; A.rep_magic = 0; A.n_in = 0; A.rep_hist = 0;
    if ((rc = fir_careful(h, &A.cf))) return rc;
    int64_t grid = 2 * (int64_t)ctx().num_cus;
    const int reserve_wgs = opt().ols_reserve;
    if (reserve_wgs > 0 && grid >= 4 * (int64_t)reserve_wgs) grid -= reserve_wgs;
    if (grid > ntiles) grid = ntiles;
    if (xr) {
        // (L = 2 is one pair: "row 0" of the rows form IS the output, written with the plain complex filter's full-width stores.  The
        // instantiation that also keeps H in registers -- UP = false -- compiles to 18 spilled registers with the real-input loads.)
        if (L == 2 && A.up_pitch == 0) A.up_pitch = 1;
        hipLaunchKernelGGL((ols_tile_kernel<false, false, true, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
    } else if (dec > 1) {
        if (real) hipLaunchKernelGGL((ols_tile_kernel<true, true, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((ols_tile_kernel<false, true, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
    } else {
        if (real) hipLaunchKernelGGL((ols_tile_kernel<true, false, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((ols_tile_kernel<false, false, true>), dim3((unsigned)grid), dim3(256), 0, s, A);
    }
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

// multirate_FIR.up, even L, at most 4097 taps in all: tiles of the OUTPUT, the zero-stuffed tile's spectrum from its non-zero columns (ols_rep_kernel)
bool fir_ols_rep_supported(const FirHandle *h, int L)
{
    if (L < 2 || L % 2 || L > 4096 || !opt().fir_up_rep) return false;
    return fir_ols_supported(h);
}

int fir_ols_rep_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, int L, void *y, hipStream_t s)
{
    note_path("fir_ols_rep");
    if (n <= 0) return SKDSP_OK;
    SK_CHECK(fir_ols_rep_supported(h, L), SKDSP_ERR_UNSUPPORTED, "fir_ols_rep: needs complex64 (or float32 with real taps), an even L and 2..4097 taps");
    OlsPlan *p = nullptr;
    int rc = up_plan(h, L, 3, &p);
    if (rc) return rc;
    const bool real = h->dtype == SKDSP_F32;
    const int lf = L % 16 == 0 ? 16 : (L % 8 == 0 ? 8 : (L % 4 == 0 ? 4 : 2));
    OlsArgs A;
    A.x = (const cf *)x;
    A.y = (cf *)y;
    A.n = n * L;                  // (the rate the tiles live at)
    A.n_hist = n_hist * L;
    A.T1 = p->T1; A.T2 = p->T2; A.Hp = p->Hp;
    A.ov = p->ov; A.V = p->V; A.a0 = p->ov / 512; A.keep = opt().ols_keep_overlap ? A.a0 : 0;
    A.aligned = ((((uintptr_t)x) | ((uintptr_t)y)) & (real ? 3 : 7)) == 0;
    int64_t ntiles = (A.n + p->V - 1) / p->V;
    if (real) ntiles = (ntiles + 1) / 2;
    SK_CHECK(ntiles < (int64_t)1 << 31, SKDSP_ERR_BADARG, "fir_ols_rep: too many tiles");
    A.ntiles = ntiles;
    A.dec = 1; A.dec_magic = 0; A.n_keep = A.n;
    A.up = 1; A.up_pitch = 0; A.up_sb = A.up_pbs = A.up_pb0 = 0;
    A.halo_flag = nullptr; A.halo_seq = 0; A.halo_err = nullptr; A.halo_spins = 0;
    A.rep_L = L; A.rep_lr = L / lf;
    A.rep_magic = A.rep_lr > 1 ? (unsigned)((((unsigned long long)1 << 32) + A.rep_lr - 1) / A.rep_lr) : 0u;
    A.n_in = n; A.rep_hist = n_hist;
    if ((rc = fir_careful(h, &A.cf))) return rc;
    int64_t grid = 2 * (int64_t)ctx().num_cus;
    const int reserve_wgs = opt().ols_reserve;
    if (reserve_wgs > 0 && grid >= 4 * (int64_t)reserve_wgs) grid -= reserve_wgs;
    if (grid > ntiles) grid = ntiles;
#define SK_REP(LFV)                                                                                                  \
    if (real) hipLaunchKernelGGL((ols_rep_kernel<true, LFV>), dim3((unsigned)grid), dim3(256), 0, s, A);             \
    else hipLaunchKernelGGL((ols_rep_kernel<false, LFV>), dim3((unsigned)grid), dim3(256), 0, s, A)
    switch (lf) {
    case 2: SK_REP(2); break;
    case 4: SK_REP(4); break;
    case 8: SK_REP(8); break;
    default: SK_REP(16); break;
    }
#undef SK_REP
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

int fir_ols_up_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, int L, void *y, hipStream_t s, int dec, int64_t rows_pitch, int paired)
{
    note_path("fir_ols_up");
    if (n <= 0) return SKDSP_OK;
    SK_CHECK(dec >= 1 && dec <= 4096 && (dec == 1 || L <= 64), SKDSP_ERR_UNSUPPORTED, "fir_ols_up: L / M = %d / %d (the fused L / M store takes L <= 64, M <= 4096)", L, dec);
    SK_CHECK(fir_ols_up_supported(h, L), SKDSP_ERR_UNSUPPORTED, "fir_ols_up: needs complex64 (or float32 with real taps), 2 <= L <= 256, 2..4097 taps per phase");
    if (!paired) return up_walk(h, 0, x, n, n_hist, L, y, s, dec, rows_pitch);
    // (rows_pitch of a paired launch counts 8-byte elements: the caller weaves L / 2 rows of pairs)
    SK_CHECK(fir_ols_up_pairs(h, L, dec, y) && (L % 2 == 0 || rows_pitch == 0), SKDSP_ERR_BADARG,
             "fir_ols_up: phases in pairs need float32, real taps, no decimation, and an 8-byte aligned destination (even L) or the strided form (odd L, 7 .. 13)");
    int rc = up_walk(h, 1, x, n, n_hist, L, y, s, 1, rows_pitch);
    if (rc || L % 2 == 0) return rc;
    return up_walk(h, 2, x, n, n_hist, L, y, s, 1, 0);   // the odd L's last phase
}

}  // namespace skdsp
