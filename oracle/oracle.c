/*
 * oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * CPU restatement, in plain C / float64, of the arithmetic that the reference's
 * streaming-filter hot path executes.  The reference (scikit-dsp-comm, pure
 * Python) delegates that arithmetic to third-party SciPy/NumPy, which is not
 * under /root/reference (requirements.txt:1-3 pins only lower bounds; the dev
 * container has SciPy 1.15.3 / NumPy 2.2.6).  The published algorithms restated
 * here, with the reference call sites they serve:
 *
 *   orc_fir_*      scipy.signal.lfilter(b,[1],x) FIR branch == np.convolve(b,x)[:N]
 *                  -> multirate_helper.py:108,117,125
 *   orc_lfilter_*  scipy.signal.lfilter(b,a,x) == direct-form II transposed,
 *                  a[0]-normalised, zero initial state
 *                  -> multirate_helper.py:74,81 ; sigsys.py:2972-2984,3015-3027
 *   orc_sosfilt_*  scipy.signal.sosfilt(sos,x): per sample, per section DF2T
 *                  -> multirate_helper.py:173,182,190
 *
 * Parity is pinned by tests/golden/ (.npz fixtures captured from the real reference by
 * tests/golden/gen_golden.py) -- see tests/test_oracle_golden.py.
 *
 * Complex vectors are interleaved (re,im) doubles.  "hist" = samples that
 * precede x[0] (hist[nh-1] is x[-1]); NULL / nh==0 means zero initial state,
 * which is what every reference call uses.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Blocked kernel shared by the real-tap variants.  Interleaved complex data with
 * real taps is just a real FIR on the 2N-double array with tap stride `st`=2
 * (st=1 for real data).  For each block of outputs the taps are applied in
 * k = 0..P-1 order, so every y[i] is the plain sequential sum (same order as the
 * naive loop); the inner loop runs over outputs and vectorises without
 * re-association (compile with -ffp-contract=off, no -ffast-math). */
#define ORC_BLK 1024
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define ORC_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define ORC_CLONES
#endif

ORC_CLONES
static void fir_block_f64(const double *b, int P, int st, const double *x, int64_t j0, int64_t j1, double *y)
{
    /* outputs y[j], j in [j0,j1), all with j - st*(P-1) >= 0 */
    for (int64_t jb = j0; jb < j1; jb += ORC_BLK) {
        int64_t len = (j1 - jb) < ORC_BLK ? (j1 - jb) : ORC_BLK;
        double acc[ORC_BLK];
        for (int64_t t = 0; t < len; ++t) acc[t] = 0.0;
        for (int k = 0; k < P; ++k) {
            const double bk = b[k];
            const double *xp = x + jb - (int64_t)st * k;
            for (int64_t t = 0; t < len; ++t) acc[t] += bk * xp[t];
        }
        for (int64_t t = 0; t < len; ++t) y[jb + t] = acc[t];
    }
}

ORC_CLONES
static void fir_block_f32in(const double *b, int P, int st, const float *x, int64_t j0, int64_t j1, double *y)
{
    for (int64_t jb = j0; jb < j1; jb += ORC_BLK) {
        int64_t len = (j1 - jb) < ORC_BLK ? (j1 - jb) : ORC_BLK;
        double acc[ORC_BLK];
        for (int64_t t = 0; t < len; ++t) acc[t] = 0.0;
        for (int k = 0; k < P; ++k) {
            const double bk = b[k];
            const float *xp = x + jb - (int64_t)st * k;
            for (int64_t t = 0; t < len; ++t) acc[t] += bk * (double)xp[t];
        }
        for (int64_t t = 0; t < len; ++t) y[jb + t] = acc[t];
    }
}

static void fir_real_taps(const double *b, int P, int st, const double *x, int64_t n, const double *hist,
                          int64_t nh, double *y, int nthreads)
{
    /* head: outputs that reach before x[0] (uses hist or zeros) */
    int64_t head = (int64_t)(P - 1) < n ? (int64_t)(P - 1) : n;
    for (int64_t i = 0; i < head; ++i) {
        for (int c = 0; c < st; ++c) {
            double acc = 0.0;
            for (int k = 0; k < P; ++k) {
                int64_t j = i - k;
                double xv = 0.0;
                if (j >= 0) xv = x[st * j + c];
                else if (hist && -j <= nh) xv = hist[st * (nh + j) + c];
                acc += b[k] * xv;
            }
            y[st * i + c] = acc;
        }
    }
    if (n <= head) return;
    int64_t j0 = (int64_t)st * head, j1 = (int64_t)st * n;
    int nt = nthreads > 0 ? nthreads : 1;
    (void)nt;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt)
#endif
    for (int64_t jb = j0; jb < j1; jb += 64 * ORC_BLK) {
        int64_t je = jb + 64 * ORC_BLK < j1 ? jb + 64 * ORC_BLK : j1;
        fir_block_f64(b, P, st, x, jb, je, y);
    }
}

/* real taps, real data: y[n] = sum_k b[k] x[n-k], n in [0,N) */
void orc_fir_rr(const double *b, int P, const double *x, int64_t n, const double *hist, int64_t nh,
                double *y, int nthreads)
{
    fir_real_taps(b, P, 1, x, n, hist, nh, y, nthreads);
}

/* real taps, complex data (interleaved) */
void orc_fir_rc(const double *b, int P, const double *x, int64_t n, const double *hist, int64_t nh,
                double *y, int nthreads)
{
    fir_real_taps(b, P, 2, x, n, hist, nh, y, nthreads);
}

/* complex taps (interleaved), complex data */
void orc_fir_cc(const double *b, int P, const double *x, int64_t n, const double *hist, int64_t nh,
                double *y, int nthreads)
{
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        double ar = 0.0, ai = 0.0;
        for (int k = 0; k < P; ++k) {
            int64_t j = i - k;
            double xr = 0.0, xi = 0.0;
            if (j >= 0) { xr = x[2 * j]; xi = x[2 * j + 1]; }
            else if (hist && -j <= nh) { xr = hist[2 * (nh + j)]; xi = hist[2 * (nh + j) + 1]; }
            ar += b[2 * k] * xr - b[2 * k + 1] * xi;
            ai += b[2 * k] * xi + b[2 * k + 1] * xr;
        }
        y[2 * i] = ar;
        y[2 * i + 1] = ai;
    }
}

/* float32 / complex64 in, float64 / complex128 out: the exact shape of the
 * reference call on the benchmark configs (lfilter promotes to double because
 * a=[1] is an int64 array).  Used as bench.py's cpu_baseline ("port"). */
static void fir_real_taps_f32in(const double *b, int P, int st, const float *x, int64_t n, double *y, int nthreads)
{
    int64_t head = (int64_t)(P - 1) < n ? (int64_t)(P - 1) : n;
    for (int64_t i = 0; i < head; ++i)
        for (int c = 0; c < st; ++c) {
            double acc = 0.0;
            for (int k = 0; k <= i; ++k) acc += b[k] * (double)x[st * (i - k) + c];
            y[st * i + c] = acc;
        }
    if (n <= head) return;
    int64_t j0 = (int64_t)st * head, j1 = (int64_t)st * n;
    int nt = nthreads > 0 ? nthreads : 1;
    (void)nt;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt)
#endif
    for (int64_t jb = j0; jb < j1; jb += 64 * ORC_BLK) {
        int64_t je = jb + 64 * ORC_BLK < j1 ? jb + 64 * ORC_BLK : j1;
        fir_block_f32in(b, P, st, x, jb, je, y);
    }
}

void orc_fir_rc_f32in(const double *b, int P, const float *x, int64_t n, double *y, int nthreads)
{
    fir_real_taps_f32in(b, P, 2, x, n, y, nthreads);
}

void orc_fir_rr_f32in(const double *b, int P, const float *x, int64_t n, double *y, int nthreads)
{
    fir_real_taps_f32in(b, P, 1, x, n, y, nthreads);
}

/* scipy.signal.sosfilt restated: sample-major loop, inner loop over sections,
 * DF2T per biquad, zero initial state unless zi given (nsec x 2, updated in place).
 * ncomp = 1 (real) or 2 (interleaved complex; real sos acts per component). */
int orc_sosfilt(const double *sos, int nsec, const double *x, int64_t n, int ncomp, double *zi, double *y)
{
    for (int s = 0; s < nsec; ++s)
        if (sos[6 * s + 3] != 1.0) return -1; /* scipy: sos[:, 3] should be all ones */
    for (int c = 0; c < ncomp; ++c) {
        double z[2 * 64];
        if (nsec > 64) return -2;
        for (int s = 0; s < 2 * nsec; ++s) z[s] = zi ? zi[(size_t)c * 2 * nsec + s] : 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double xc = x[(size_t)i * ncomp + c];
            for (int s = 0; s < nsec; ++s) {
                const double *q = sos + 6 * s;
                double xn = xc;
                xc = q[0] * xn + z[2 * s];
                z[2 * s] = q[1] * xn - q[4] * xc + z[2 * s + 1];
                z[2 * s + 1] = q[2] * xn - q[5] * xc;
            }
            y[(size_t)i * ncomp + c] = xc;
        }
        if (zi) for (int s = 0; s < 2 * nsec; ++s) zi[(size_t)c * 2 * nsec + s] = z[s];
    }
    return 0;
}

/* float32-in / float64-out variant (reference cfg 4 shape) for the cpu_baseline */
int orc_sosfilt_f32in(const double *sos, int nsec, const float *x, int64_t n, double *y)
{
    double z[2 * 64];
    if (nsec > 64) return -2;
    for (int s = 0; s < 2 * nsec; ++s) z[s] = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double xc = (double)x[i];
        for (int s = 0; s < nsec; ++s) {
            const double *q = sos + 6 * s;
            double xn = xc;
            xc = q[0] * xn + z[2 * s];
            z[2 * s] = q[1] * xn - q[4] * xc + z[2 * s + 1];
            z[2 * s + 1] = q[2] * xn - q[5] * xc;
        }
        y[i] = xc;
    }
    return 0;
}

/* scipy.signal.lfilter(b,a,x) restated (_sigtools._linear_filter): coefficients
 * divided by a[0], both padded to K=max(nb,na), direct-form II transposed:
 *   y = z0 + b0*x ; z[k-1] = z[k] + b[k]*x - a[k]*y ; z[K-2] = b[K-1]*x - a[K-1]*y */
int orc_lfilter(const double *b, int nb, const double *a, int na, const double *x, int64_t n, int ncomp, double *y)
{
    int K = nb > na ? nb : na;
    if (K > 128 || na < 1 || a[0] == 0.0) return -1;
    double bb[128], aa[128], z[128];
    for (int k = 0; k < K; ++k) {
        bb[k] = (k < nb ? b[k] : 0.0) / a[0];
        aa[k] = (k < na ? a[k] : 0.0) / a[0];
    }
    for (int c = 0; c < ncomp; ++c) {
        for (int k = 0; k < K; ++k) z[k] = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double xn = x[(size_t)i * ncomp + c];
            double yn;
            if (K > 1) {
                yn = z[0] + bb[0] * xn;
                for (int k = 1; k < K - 1; ++k) z[k - 1] = z[k] + xn * bb[k] - yn * aa[k];
                z[K - 2] = xn * bb[K - 1] - yn * aa[K - 1];
            } else {
                yn = xn * bb[0];
            }
            y[(size_t)i * ncomp + c] = yn;
        }
    }
    return 0;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
