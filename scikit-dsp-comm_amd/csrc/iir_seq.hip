// iir_seq.hip -- the reference's own recursion, sample by sample, for cascades no scan can be trusted with.
//
// scipy.signal.sosfilt(sos, x) (multirate_helper.py:173, :181, :190) runs every sample through every section in turn (direct form II
// transposed).  The scan kernels of this library (iir_par.hip, iir_fused.hip, iir_scan.hip) compute the same outputs from chunk
// transitions instead; for well-conditioned cascades the two agree to the rounding of float64, but a 40th-order Chebyshev cascade is good
// to 1e-6 of its output in float64 AT BEST -- two evaluations of the reference itself, sections in another order, lie that far apart --
// and the scans lose another 30 - 400 x that spread on such designs (measured: profiles/r05/iir_illcond.txt).  Handles whose cascade
// shows a float64 spread that the scans would lift above the contract (capi.hip: the probe at creation) therefore run HERE: the recursion
// itself, in the reference's operation order, without fused multiply-adds -- bit for bit what scipy computes for float64 signals.
//
// One wave per row (a complex signal is two rows).  Lane s holds section s; the samples move down the lanes in BLOCKS of kSeqB: at block
// step t lane s runs block t - s through its section -- kSeqB samples one after the other, the section's state in registers -- and hands
// the block to lane s + 1 by one-lane DPP shifts.  Round 5 handed over sample by sample: per sample a broadcast of the input to lane 0
// (two v_readlane + two selects), of the last section's result (two more) and its placement in an output register, five range tests -- 35
// vector instructions around the 9 of the recursion, ~150 clocks per sample on the one wave that issues them (5 - 6 MSamples/s on an
// idle, down-clocked chip).  In blocks the hand-over is two DPP moves per sample and nothing else: lane 0's next block arrives by ITS
// loads into the registers the shift leaves to it (`old` of the DPP move), the last section's lane stores ITS block, and the range tests
// are one per block.  13 - 14 instructions per sample.  64 sections per pass; longer cascades run pass after pass through a float64
// buffer of the handle (so float32 signals are rounded once, at the end, and a decimating call keeps the full rate between passes).
#include "skdsp_internal.hpp"

namespace skdsp {

namespace {

constexpr int kSeqB = 16;   // samples per block

struct SeqArgs {
    const void *x;
    void *y;
    int64_t n;                    // samples per row
    int64_t x_stride, y_stride;   // elements between rows
    const double *coef;           // [ns][5]: b0, b1, b2, a1, a2 of this pass's sections (the caller's factorisation)
    int ns;                       // sections of this pass (1 .. 64)
    double *state;                // [rows][2 ns]: DF2T states in, states out (scipy's zi / zf coordinates); null = from rest, not returned
    int dec;                      // > 1 (last pass): only y[k dec] is stored, at y[k]
};

// lanes 1 .. 63: v of lane - 1; lane 0: `keep` (its own next input)
__device__ __forceinline__ double lane_shr1_or(double keep, double v)
{
    const long long b = __double_as_longlong(v), k = __double_as_longlong(keep);
    const int lo2 = __builtin_amdgcn_update_dpp((int)k, (int)b, 0x138, 0xf, 0xf, false);           // wave_shr:1 (lane 0 has no source: keeps `old`)
    const int hi2 = __builtin_amdgcn_update_dpp((int)(k >> 32), (int)(b >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi2 << 32) | (unsigned)lo2);
}

template <typename IN, typename OUT>
__global__ __launch_bounds__(64) void iir_seq_kernel(SeqArgs a)
{
#pragma clang fp contract(off)   // (the reference rounds every product and every sum: no fused multiply-adds here)
    constexpr int B = kSeqB;
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    const IN *x = reinterpret_cast<const IN *>(a.x) + row * a.x_stride;
    OUT *y = reinterpret_cast<OUT *>(a.y) + row * a.y_stride;
    const bool mine = lane < a.ns;
    double b0 = 0, b1 = 0, b2 = 0, a1 = 0, a2 = 0, z0 = 0, z1 = 0;
    if (mine) {
        const double *c = a.coef + 5 * lane;
        b0 = c[0]; b1 = c[1]; b2 = c[2]; a1 = c[3]; a2 = c[4];
        if (a.state) {
            z0 = a.state[row * 2 * a.ns + 2 * lane];
            z1 = a.state[row * 2 * a.ns + 2 * lane + 1];
        }
    }
    const int last = a.ns - 1;
    const int64_t nblk = (a.n + B - 1) / B;
    const int64_t nsteps = nblk + last;
    // lane 0's block `blk` of the row, as doubles (beyond the row: zeros; the last, partial block element by element -- nothing is read past x[n - 1])
    auto fetch = [&](int64_t blk, double (&v)[B]) {
        const int64_t i0 = blk * B;
        if (i0 + B <= a.n) {
#pragma unroll
            for (int j = 0; j < B; ++j) v[j] = (double)x[i0 + j];
        } else {
#pragma unroll
            for (int j = 0; j < B; ++j) v[j] = i0 + j < a.n ? (double)x[i0 + j] : 0.0;
        }
    };
    double w[B];       // the block this lane works on: inputs, then in place its section's outputs
    double nx[B];      // lane 0: the next block of the row (requested a block step ahead -- which also makes in-place calls safe: a block is read a step
                       // before the first section starts on it, and written when the last section is through with it)
#pragma unroll
    for (int j = 0; j < B; ++j) { w[j] = 0.0; nx[j] = 0.0; }
    if (lane == 0) fetch(0, w);
    // (the reference's statement order: x_c = b0 x_n + z0; z0 = b1 x_n - a1 x_c + z1; z1 = b2 x_n - a2 x_c.  Plain operators under the pragma above: the
    // __dmul_rn / __dadd_rn of the HIP headers are ordinary inline functions whose products and sums carry their own contraction licence)
    auto section = [&](double in) -> double {
        const double xc = b0 * in + z0;
        z0 = b1 * in - a1 * xc + z1;
        z1 = b2 * in - a2 * xc;
        return xc;
    };
#pragma unroll 1
    for (int64_t t = 0; t < nsteps; ++t) {
        if (lane == 0 && t + 1 < nblk) fetch(t + 1, nx);
        const int64_t blk = t - lane;                       // this lane's block
        const int64_t left = a.n - blk * B;                 // its samples, if it is a block of the row
        const int cnt = !mine || blk < 0 || left <= 0 ? 0 : (left >= B ? B : (int)left);
        if (cnt == B) {                                     // (the steady state: every lane of the wave, one test per block)
#pragma unroll
            for (int j = 0; j < B; ++j) w[j] = section(w[j]);
        } else if (cnt > 0) {                               // the row's last, partial block
#pragma unroll
            for (int j = 0; j < B; ++j)
                if (j < cnt) w[j] = section(w[j]);
        }
        if (lane == last && cnt > 0) {                      // the last section's lane stores its block
            const int64_t i0 = blk * B;
            if (a.dec <= 1) {
                if (cnt == B) {
#pragma unroll
                    for (int j = 0; j < B; ++j) y[i0 + j] = (OUT)w[j];
                } else {
#pragma unroll
                    for (int j = 0; j < B; ++j)
                        if (j < cnt) y[i0 + j] = (OUT)w[j];
                }
            } else {
                const int64_t n_out = a.n / a.dec;
                int64_t o = (i0 + a.dec - 1) / a.dec;       // first kept sample of the block: o dec >= i0
                int jn = (int)(o * a.dec - i0);
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    if (j == jn && j < cnt) {
                        if (o < n_out) y[o] = (OUT)w[j];
                        ++o;
                        jn += a.dec;
                    }
                }
            }
        }
        // hand the blocks down: lane s + 1 receives lane s's results, lane 0 its next block of the row
#pragma unroll
        for (int j = 0; j < B; ++j) w[j] = lane_shr1_or(nx[j], w[j]);
    }
    if (mine && a.state) {
        a.state[row * 2 * a.ns + 2 * lane] = z0;
        a.state[row * 2 * a.ns + 2 * lane + 1] = z1;
    }
}

template <typename IN, typename OUT> static void seq_launch_one(int nrow, hipStream_t s, const SeqArgs &a)
{
    hipLaunchKernelGGL((iir_seq_kernel<IN, OUT>), dim3((unsigned)nrow), dim3(64), 0, s, a);
}

}  // namespace

// x / y: real planar rows in the handle's precision (complex callers pass two rows); y may alias x.  zi_host / zf_host: [rows][2 nsec]
// in scipy's coordinates, null = from rest / not wanted.  dec > 1: one real row, y receives n / dec samples.
int iir_seq_launch(IirHandle *h, const void *x, int64_t n, int nrow, int64_t x_stride, int64_t y_stride, void *y, hipStream_t s,
                   const double *zi_host, double *zf_host, int dec)
{
    if (n <= 0 || nrow <= 0) return SKDSP_OK;
    SK_CHECK(h->order == 2 && !h->seq_coef.empty(), SKDSP_ERR_UNSUPPORTED, "iir_seq: second-order sections only");
    SK_CHECK(dec <= 1 || nrow == 1, SKDSP_ERR_UNSUPPORTED, "iir_seq: the decimating store takes one row");
    note_path("iir_seq");
    const int nsec = h->nsec;
    if (!h->seq_coef_dev) {
        SK_HIP(hipMalloc(&h->seq_coef_dev, h->seq_coef.size() * 8));
        SK_HIP(hipMemcpy(h->seq_coef_dev, h->seq_coef.data(), h->seq_coef.size() * 8, hipMemcpyHostToDevice));
    }
    const bool dbl = dtype_double(h->dtype);
    const bool with_state = zi_host || zf_host;
    const int npass = (nsec + 63) / 64;
    // more than 64 sections: the passes meet in a float64 buffer of the handle, [rows][n] -- the signal between two passes is the reference's own float64
    // intermediate (a float32 handle rounds once, at the last pass), and a decimating call keeps the full rate until its last pass
    double *mid = nullptr;
    if (npass > 1) {
        const size_t need = (size_t)nrow * (size_t)n * 8 + 256;
        if (need > h->group_tmp_bytes) {
            if (h->group_tmp) {
                SK_HIP(hipStreamSynchronize(s));
                SK_HIP(hipFree(h->group_tmp));
                h->group_tmp = nullptr; h->group_tmp_bytes = 0;
            }
            SK_HIP(hipMalloc(&h->group_tmp, need));
            h->group_tmp_bytes = need;
        }
        mid = static_cast<double *>(h->group_tmp);
    }
    std::vector<double> st;   // per pass [rows][2 ns], packed pass after pass
    double *st_dev = nullptr;
    if (with_state) {
        st.assign((size_t)nrow * 2 * nsec, 0.0);
        size_t at = 0;
        for (int p = 0; p < npass; ++p) {
            const int s0 = 64 * p, ns = std::min(64, nsec - s0);
            for (int r = 0; r < nrow; ++r)
                for (int d = 0; d < 2 * ns; ++d) st[at + (size_t)r * 2 * ns + d] = zi_host ? zi_host[(size_t)r * 2 * nsec + 2 * s0 + d] : 0.0;
            at += (size_t)nrow * 2 * ns;
        }
        SK_HIP(hipMalloc((void **)&st_dev, st.size() * 8));
        SK_HIP(hipMemcpyAsync(st_dev, st.data(), st.size() * 8, hipMemcpyHostToDevice, s));
    }
    size_t at = 0;
    for (int p = 0; p < npass; ++p) {
        const int s0 = 64 * p, ns = std::min(64, nsec - s0);
        const bool first = p == 0, lastp = p + 1 == npass;
        SeqArgs a;
        a.x = first ? x : mid;
        a.y = lastp ? y : mid;
        a.n = n;
        a.x_stride = first ? x_stride : n;
        a.y_stride = lastp ? y_stride : n;
        a.coef = (const double *)h->seq_coef_dev + 5 * s0;
        a.ns = ns;
        a.state = with_state ? st_dev + at : nullptr;
        a.dec = lastp && dec > 1 ? dec : 1;
        const bool in_d = dbl || !first, out_d = dbl || !lastp;   // (the buffer between passes is float64 whatever the handle)
        if (in_d && out_d) seq_launch_one<double, double>(nrow, s, a);
        else if (in_d) seq_launch_one<double, float>(nrow, s, a);
        else if (out_d) seq_launch_one<float, double>(nrow, s, a);
        else seq_launch_one<float, float>(nrow, s, a);
        at += (size_t)nrow * 2 * ns;
    }
    SK_HIP(hipGetLastError());
    if (with_state) {
        SK_HIP(hipMemcpyAsync(st.data(), st_dev, st.size() * 8, hipMemcpyDeviceToHost, s));
        SK_HIP(hipStreamSynchronize(s));
        (void)hipFree(st_dev);
        if (zf_host) {
            size_t a2 = 0;
            for (int p = 0; p < npass; ++p) {
                const int s0 = 64 * p, ns = std::min(64, nsec - s0);
                for (int r = 0; r < nrow; ++r)
                    for (int d = 0; d < 2 * ns; ++d) zf_host[(size_t)r * 2 * nsec + 2 * s0 + d] = st[a2 + (size_t)r * 2 * ns + d];
                a2 += (size_t)nrow * 2 * ns;
            }
        }
    }
    return SKDSP_OK;
}

}  // namespace skdsp
