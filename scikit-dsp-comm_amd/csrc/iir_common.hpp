// iir_common.hpp -- declarations shared by the IIR scan translation units (iir_scan.hip: K1 / carries / K3 and the
// host side; iir_fused.hip: the single-pass scan).  gfx950 only.
#pragma once
#include "skdsp_internal.hpp"
#include <type_traits>
#include <cmath>
#include <cstring>

namespace skdsp {

constexpr int kIirThreads = 256;
constexpr int kPiece = 32;          // samples per thread per staged piece
constexpr int kMaxPairs = 12288;    // K2 capacity: workgroups x state dimension (one 96 KiB LDS image)
constexpr int kMaxW = 512;          // workgroups (of 256 chunks) per vector (1024 measured slower: more scan, same occupancy)
constexpr int kPowers = 19;         // M^(2^l), l = 0..18

struct IirPlan {
    int nsec, order, D;
    // per call geometry is recomputed; matrix powers are cached per chunk length T
    int64_t cached_T = -1;
    double *pw_dev = nullptr;    // kPowers matrices M^(2^l), each D x D row-major
    double *pwa_dev = nullptr;   // M^(2^l), l = 0..7, zero-padded to 16 x 16, as MFMA A operands [l][k / 4][64] (D <= 16)
    double *lb_dev = nullptr;    // look-back matrices (M^256)^k, k = 1..7
    double *lbk_dev = nullptr;   // chunk look-back powers M^k, k = 0..31, lane-contiguous (aggregate-free mode)
    double *gt_dev = nullptr;    // G = [A^(T-1-k) b]_k in MFMA A-operand order: [T/4][64], grown on demand
    size_t gt_cap = 0;
    int64_t gt_T = -1;           // chunk length the table was built for (-1: none)
    int n_lb = 0;                // terms of the in-kernel carry look-back (0 = use the K2 scan)
    int n_lv = kPowers;          // first l with max|M^(2^l)| < 1e-30 (chunk-level scan depth that matters)
    double *state_dev = nullptr; // [2][2][D]: zi and zf for up to two planes
    double *v_dev = nullptr;     // [D][J] chunk end states (SoA), capacity below
    double *agg_dev = nullptr;   // [2][kMaxW][D] workgroup aggregates / carries (ping-pong) + carry
    size_t v_cap = 0;
    std::vector<double> A_host;
    // single-pass scan (iir_fused_kernel)
    int fused_state = 0;                   // 0 untested, 1 applicable, -1 not (the segment transition does not vanish)
    int fused_state_c = 0;                 // the same for interleaved complex signals (chunks half as long)
    int fused_nlv = 0, fused_nlv_c = 0;    // scan levels (n_lv) at the single-pass chunk length, for the path policy
    unsigned long long *lbg_dev = nullptr;  // look-back granules [batch][nseg][32]
    size_t lbg_cap = 0;
    unsigned long long *ticket_dev = nullptr;  // [2] segment dispensers, monotonic
    unsigned long long ticket_count[2] = {0, 0};  // their values (a launch adds nseg to each dispenser it uses)
    unsigned epoch = 0;
};

template <int NSEC, int ORD> struct Coef { double c[NSEC * (2 * ORD + 1)]; };

// one sample through the cascade; z = DF2T delay lines of every section
// UNIT (biquads): sections 1.. have b0 = b2 = 1 (IirHandle::unit_tail): y = x + z0, z0 = b1 x - a1 y + z1,
// z1 = x - a2 y -- 4 flops and 3 coefficients instead of 5 and 5.  The coefficients live in SGPRs; the general
// form of an 8-biquad cascade needs 80 of them, more than a wave has, and hipcc then re-reads ~9 spilled
// coefficients per sample from VGPR lanes (1100 v_readlane next to 3100 FP64 instructions in K3).
template <int NSEC, int ORD, bool UNIT = false>
__device__ __forceinline__ double cascade_step(const Coef<NSEC, ORD> &cf, double (&z)[NSEC * ORD], double x)
{
#pragma unroll
    for (int s = 0; s < NSEC; ++s) {
        const double *c = cf.c + s * (2 * ORD + 1);
        const double xn = x;
        if (UNIT && ORD == 2 && s >= 1) {
            const double yu = xn + z[2 * s];
            z[2 * s] = fma(c[1], xn, fma(-c[3], yu, z[2 * s + 1]));
            z[2 * s + 1] = fma(-c[4], yu, xn);
            x = yu;
            continue;
        }
        const double yv = fma(c[0], xn, z[s * ORD]);
#pragma unroll
        for (int k = 1; k < ORD; ++k) z[s * ORD + k - 1] = fma(c[k], xn, fma(-c[ORD + k], yv, z[s * ORD + k]));
        z[s * ORD + ORD - 1] = fma(c[ORD], xn, -c[2 * ORD] * yv);
        x = yv;
    }
    return x;
}

// out += Mat * in   (Mat uniform, row-major D x D, read through the scalar cache).  Every power
// of a cascade's transition matrix is block lower-triangular (section s never sees the state of a
// later section), so row i stops at the end of its own ORD-wide block: 144 instead of 256 fma for
// 8 biquads.
template <int D, int ORD = D>
__device__ __forceinline__ void matvec_acc(const double *__restrict__ Mat, const double (&in)[D], double (&out)[D])
{
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double acc = out[i];
#pragma unroll
        for (int j = 0; j < (i / ORD + 1) * ORD; ++j) acc = fma(Mat[i * D + j], in[j], acc);
        out[i] = acc;
    }
}

template <typename IO> struct Stage;
template <> struct Stage<float> {
    static constexpr int pitch = 36;       // floats per row (144 B)
    static constexpr int segs = 8;         // 16-byte segments per 32-sample row piece
    static constexpr int per_thread = 8;   // segments staged per thread per piece
    static constexpr int elems = 4;        // samples per 16-byte segment
};
template <> struct Stage<double> {
    static constexpr int pitch = 34;       // doubles per row (272 B)
    static constexpr int segs = 16;
    static constexpr int per_thread = 16;
    static constexpr int elems = 2;
};

typedef double v4d_t __attribute__((ext_vector_type(4)));

// LDS hand-over between the lanes of ONE wave: the hardware serves a wave's LDS instructions in order; the fence keeps the
// compiler from moving a lane's reads above its writes
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}


// single-pass scan (iir_fused.hip); T = 128 (float) / 64 (double) samples per chunk; the plan's powers and G table
// must already be those of that chunk length (ensure_powers in iir_scan.hip)
int iir_fused_launch(IirHandle *h, const void *x, int64_t n, int nbatch, int64_t batch_stride, void *y, const double *zi_dev,
                     double *zf_dev, hipStream_t s, int dec = 1,
                     int interleaved = 0);  // 1: x / y interleaved complex (chunks of 64 / 32 complex samples), nbatch = 2

}  // namespace skdsp
