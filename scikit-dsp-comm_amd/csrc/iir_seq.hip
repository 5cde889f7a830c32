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
// One wave per row (a complex signal is two rows).  Lane s holds section s; the samples move down the lanes one per step (a systolic
// pipeline: lane s works on sample t - s at step t), handed over by a one-lane DPP shift.  64 sections per pass; longer cascades run pass
// after pass in place.  About 40 - 60 clocks per step: 2^26 samples take seconds -- the reference's own speed on a host core, which is
// what such a filter costs; the Python layer logs a WARNING when it makes such a handle.
#include "skdsp_internal.hpp"

namespace skdsp {

namespace {

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

// (v of lane - 1; lane 0 keeps its own)
__device__ __forceinline__ double lane_shr1(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = (int)b, hi = (int)(b >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);   // wave_shr:1
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi2 << 32) | (unsigned)lo2);
}

// lane `src` (wave-uniform) of v, to every lane
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <typename IO>
__global__ __launch_bounds__(64) void iir_seq_kernel(SeqArgs a)
{
#pragma clang fp contract(off)   // (the reference rounds every product and every sum: no fused multiply-adds here)
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    const IO *x = reinterpret_cast<const IO *>(a.x) + row * a.x_stride;
    IO *y = reinterpret_cast<IO *>(a.y) + row * a.y_stride;
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
    const int64_t steps = a.n + a.ns - 1;
    const int last = a.ns - 1;
    double out = 0.0;    // what this lane handed down at the previous step
    double yblk = 0.0;   // lane l: output base + l of the block the last section is working on
    // The row travels through registers: lane l holds sample t0 + l of the current block of 64 (one coalesced load per block, requested a block
    // ahead -- which also makes in-place calls safe: a block is read before any of its samples is overwritten), lane 0 picks sample t0 + j with a
    // v_readlane; the last section's results are dropped into lane il & 63 of yblk and leave as one coalesced store per 64.
    auto fetch = [&](int64_t blk) -> double {
        const int64_t i = blk * 64 + lane;
        return i < a.n ? (double)x[i] : 0.0;
    };
    double xcur = fetch(0);
#pragma unroll 1
    for (int64_t t0 = 0; t0 < steps; t0 += 64) {
        const double xnext = fetch(t0 / 64 + 1);
        // everything the 64 steps of this block test is worked out here, in 32 bits: this lane works on sample t0 - lane + j in steps
        // j in [jlo, jhi); the last section finishes sample t0 - last + j in steps [llo, lhi) and a block of 64 outputs is complete at step jf
        const int64_t i0 = t0 - lane, l0 = t0 - last;
        const int jlo = i0 >= 0 ? 0 : (i0 < -64 ? 64 : (int)-i0), jhi = a.n - i0 >= 64 ? 64 : (a.n - i0 <= 0 ? 0 : (int)(a.n - i0));
        const int llo = l0 >= 0 ? 0 : (l0 < -64 ? 64 : (int)-l0), lhi = a.n - l0 >= 64 ? 64 : (a.n - l0 <= 0 ? 0 : (int)(a.n - l0));
        const int jf = (int)((63 - (l0 & 63)) & 63);                      // (l0 + jf) & 63 == 63
        const int jend = a.n - 1 - l0 >= 0 && a.n - 1 - l0 < 64 ? (int)(a.n - 1 - l0) : -1;   // the row's last sample, if this block finishes it
        const int slot0 = (int)(l0 & 63);
        const int jmax = steps - t0 >= 64 ? 64 : (int)(steps - t0);
#pragma unroll 2
        for (int j = 0; j < jmax; ++j) {
            double in = lane_shr1(out);
            const double x0 = lane_bcast(xcur, j);
            if (lane == 0) in = x0;
            if (mine && j >= jlo && j < jhi) {
                // (the reference's statement order: x_c = b0 x_n + z0; z0 = b1 x_n - a1 x_c + z1; z1 = b2 x_n - a2 x_c.  Plain operators under the pragma
                // above: the __dmul_rn / __dadd_rn of the HIP headers are ordinary inline functions whose products and sums carry their own
                // contraction licence and were fused all the same)
                const double xc = b0 * in + z0;
                z0 = b1 * in - a1 * xc + z1;
                z1 = b2 * in - a2 * xc;
                out = xc;
            }
            if (j >= llo && j < lhi) {   // the last section has just finished sample l0 + j
                const double fin = lane_bcast(out, last);
                yblk = lane == ((slot0 + j) & 63) ? fin : yblk;
                if (j == jf || j == jend) {
                    const int64_t il = l0 + j, o = (il & ~(int64_t)63) + lane;
                    if (o <= il) {
                        if (a.dec > 1) {
                            if (o % a.dec == 0 && o / a.dec < a.n / a.dec) y[o / a.dec] = (IO)yblk;
                        } else {
                            y[o] = (IO)yblk;
                        }
                    }
                }
            }
        }
        xcur = xnext;
    }
    if (mine && a.state) {
        a.state[row * 2 * a.ns + 2 * lane] = z0;
        a.state[row * 2 * a.ns + 2 * lane + 1] = z1;
    }
}

}  // namespace

// x / y: real planar rows in the handle's precision (complex callers pass two rows); y may alias x.  zi_host / zf_host: [rows][2 nsec]
// in scipy's coordinates, null = from rest / not wanted.  dec > 1: one real row, y receives n / dec samples.
int iir_seq_launch(IirHandle *h, const void *x, int64_t n, int nrow, int64_t x_stride, int64_t y_stride, void *y, hipStream_t s,
                   const double *zi_host, double *zf_host, int dec)
{
    note_path("iir_seq");
    if (n <= 0 || nrow <= 0) return SKDSP_OK;
    SK_CHECK(h->order == 2 && !h->seq_coef.empty(), SKDSP_ERR_UNSUPPORTED, "iir_seq: second-order sections only");
    SK_CHECK(dec <= 1 || nrow == 1, SKDSP_ERR_UNSUPPORTED, "iir_seq: the decimating store takes one row");
    const int nsec = h->nsec;
    if (!h->seq_coef_dev) {
        SK_HIP(hipMalloc(&h->seq_coef_dev, h->seq_coef.size() * 8));
        SK_HIP(hipMemcpy(h->seq_coef_dev, h->seq_coef.data(), h->seq_coef.size() * 8, hipMemcpyHostToDevice));
    }
    const bool dbl = dtype_double(h->dtype);
    const bool with_state = zi_host || zf_host;
    const int npass = (nsec + 63) / 64;
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
        const bool lastp = p + 1 == npass;
        SeqArgs a;
        a.x = p == 0 ? x : y;   // (later passes filter the previous pass's result in place)
        a.y = y;
        a.n = n;
        a.x_stride = p == 0 ? x_stride : y_stride;
        a.y_stride = y_stride;
        a.coef = (const double *)h->seq_coef_dev + 5 * s0;
        a.ns = ns;
        a.state = with_state ? st_dev + at : nullptr;
        a.dec = lastp && dec > 1 ? dec : 1;
        if (npass > 1 && dec > 1 && !lastp) {
            // (a full-rate intermediate cannot live in a y of n / dec samples)
            if (st_dev) (void)hipFree(st_dev);
            SK_CHECK(false, SKDSP_ERR_UNSUPPORTED, "iir_seq: a decimating call over more than 64 sections needs a full-rate buffer (not served)");
        }
        if (dbl) hipLaunchKernelGGL((iir_seq_kernel<double>), dim3((unsigned)nrow), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((iir_seq_kernel<float>), dim3((unsigned)nrow), dim3(64), 0, s, a);
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
