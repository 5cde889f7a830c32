// capi.hip -- the extern "C" boundary of libskdsp_hip.so (see include/skdsp.h).
// Runtime context (one GPU per process, one stream), grow-only staging workspaces
// for the host-pointer entry points, handle lifetime, algorithm selection.
#include "skdsp_internal.hpp"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <complex>
#include <array>
#include <algorithm>
#include <cmath>

namespace skdsp {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    (void)hipGetLastError();
    if (e == hipErrorOutOfMemory) return SKDSP_ERR_NOMEM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return SKDSP_ERR_NODEVICE;
    return SKDSP_ERR_HIP;
}

Context &ctx()
{
    static Context c;
    return c;
}

// ---- options: environment read once, skdsp_set_option afterwards ----------------------------
namespace {
struct OptEntry { const char *name; int Options::*field; };
const OptEntry kOptTable[] = {
    {"device", &Options::device}, {"fir_algo", &Options::fir_algo}, {"dn_no_ols", &Options::dn_no_ols},
    {"fir_mm", &Options::fir_mm}, {"fir_bx", &Options::fir_bx}, {"fir_no_sw", &Options::fir_no_sw},
    {"sw_no_tile", &Options::sw_no_tile}, {"sw_no_lpt", &Options::sw_no_lpt}, {"mm_ns", &Options::mm_ns},
    {"ols_reserve", &Options::ols_reserve}, {"iir_planar", &Options::iir_planar}, {"iir_no_unit", &Options::iir_no_unit},
    {"iir_dn_full", &Options::iir_dn_full}, {"iir_no_mfma", &Options::iir_no_mfma}, {"iir_no_k1r", &Options::iir_no_k1r},
    {"k1r_wgs", &Options::k1r_wgs}, {"iir_two_pass", &Options::iir_two_pass}, {"shard_no_overlap", &Options::shard_no_overlap},
    {"shard_reserve", &Options::shard_reserve}, {"shard_two_launches", &Options::shard_two_launches},
    {"shard_self_halo", &Options::shard_self_halo}, {"dist_force_comm", &Options::dist_force_comm},
    {"host_chunk_log2", &Options::host_chunk_log2}, {"host_pipeline", &Options::host_pipeline},
};
int parse_opt(const char *name, const char *v)
{
    if (!strcmp(name, "fir_algo")) {
        if (!strcmp(v, "direct")) return SKDSP_FIR_DIRECT;
        if (!strcmp(v, "ols")) return SKDSP_FIR_OLS;
        if (!strcmp(v, "auto")) return SKDSP_FIR_AUTO;
    }
    if (!*v) return 1;  // SKDSP_X= (set, empty) switches X on
    return atoi(v);
}
Options options_from_env()
{
    Options o;
    for (const OptEntry &e : kOptTable) {
        char key[64] = "SKDSP_";
        size_t k = 6;
        for (const char *p = e.name; *p && k + 1 < sizeof(key); ++p) key[k++] = (char)toupper((unsigned char)*p);
        key[k] = 0;
        if (const char *v = getenv(key)) o.*(e.field) = parse_opt(e.name, v);
    }
    return o;
}
}  // namespace

Options &opt()
{
    static Options o = options_from_env();
    return o;
}

static int init_locked(int device)
{
    Context &c = ctx();
    if (c.ready) {
        SK_CHECK(device < 0 || device == c.device, SKDSP_ERR_BADARG,
                 "skdsp_init: already bound to device %d (one GPU per process)", c.device);
        return SKDSP_OK;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (hipGetDeviceCount -> %d, %s): the MI355X path has no CPU fallback",
                  ndev, e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        (void)hipGetLastError();
        return SKDSP_ERR_NODEVICE;
    }
    if (device < 0) device = 0;
    SK_CHECK(device < ndev, SKDSP_ERR_NODEVICE, "skdsp_init: device %d out of range (%d visible)", device, ndev);
    SK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SK_HIP(hipGetDeviceProperties(&prop, device));
    c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    SK_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    SK_HIP(hipEventCreate(&c.ev_start));
    SK_HIP(hipEventCreate(&c.ev_stop));
    c.device = device;
    c.ready = true;
    return SKDSP_OK;
}

int ensure_init()
{
    Context &c = ctx();
    if (c.ready) return SKDSP_OK;
    std::lock_guard<std::mutex> lk(c.mu);
    return init_locked(opt().device);
}

int ws_reserve(int slot, size_t bytes, void **out)
{
    Context &c = ctx();
    if (bytes > c.ws_bytes[slot]) {
        if (c.ws[slot]) {
            SK_HIP(hipStreamSynchronize(c.stream));
            SK_HIP(hipFree(c.ws[slot]));
            c.ws[slot] = nullptr;
            c.ws_bytes[slot] = 0;
        }
        size_t cap = bytes + bytes / 8 + 4096;
        SK_HIP(hipMalloc(&c.ws[slot], cap));
        c.ws_bytes[slot] = cap;
    }
    *out = c.ws[slot];
    return SKDSP_OK;
}

static inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// FIR .filter algorithm choice.  OLS needs complex64; it wins once direct form stops
// being HBM-bound (2*P FMA per c64 sample on the VALU vs ~120 flop in the FFT domain).
static int pick_fir_algo(const FirHandle *h, int64_t n)
{
    int algo = opt().fir_algo != SKDSP_FIR_AUTO ? opt().fir_algo : h->algo;
    if (algo == SKDSP_FIR_OLS && !fir_ols_supported(h)) algo = SKDSP_FIR_DIRECT;
    if (algo != SKDSP_FIR_AUTO) return algo;
    // measured crossover at 2^26 samples (same box, alternating runs): the bf16x3 matrix-pipe kernel (real taps) stays
    // ahead of overlap-save up to 3 lag blocks for complex64 (0.21 vs 0.23 ms at 81 taps; 0.234 vs 0.227 at 96) and
    // 5 for float32 (0.135 vs 0.138 ms at 145 taps)
    const int ols_from = h->taps_complex ? 48 : (h->dtype == SKDSP_C64 ? 82 : 146);
    if (fir_ols_supported(h) && h->ntaps >= ols_from && n >= 4096) return SKDSP_FIR_OLS;
    return SKDSP_FIR_DIRECT;
}

int fir_algo_for(const FirHandle *h, int64_t n) { return pick_fir_algo(h, n); }

// .dn: long filters with a modest M go through the overlap-save engine with a decimating store, which
// beats Ntaps/M direct taps per kept sample (2^24 complex64, 512 taps, M = 3: 0.163 -> 0.085 ms).  Where the
// bf16x3 matrix-pipe kernel covers the geometry it is the faster one up to ~4 M lag blocks for complex64
// (2^26: 0.14-0.20 ms against a flat 0.24) and always for float32 (0.07-0.15 against 0.26).
static int fir_dn_any(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int M, void *y_dev)
{
    bool ols = M > 1 && fir_ols_supported(h) && pick_fir_algo(h, n) == SKDSP_FIR_OLS && !opt().dn_no_ols;
    if (ols) {
        const int kb = h->algo == SKDSP_FIR_OLS ? 0 : fir_bx_blocks(h, 1, M);
        if (kb > 0) ols = h->dtype == SKDSP_C64 && kb > 4 * M;
        else ols = h->ntaps / M >= (h->dtype == SKDSP_C64 ? 24 : 64);  // the two-real-tiles store pays two divides per sample
    }
    if (ols) return fir_ols_launch(h, x_dev, n, n_hist, y_dev, ctx().stream, M);
    return fir_direct_launch(h, x_dev, n, n_hist, 1, M, n / M, y_dev, ctx().stream);
}

static int fir_filter_any(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, void *y_dev)
{
    if (pick_fir_algo(h, n) == SKDSP_FIR_OLS) return fir_ols_launch(h, x_dev, n, n_hist, y_dev, ctx().stream);
    return fir_direct_launch(h, x_dev, n, n_hist, 1, 1, n, y_dev, ctx().stream);
}

template <typename H> static H *as_handle(skdsp_handle h, int kind)
{
    HandleBase *b = reinterpret_cast<HandleBase *>(h);
    if (!b || b->kind != kind) return nullptr;
    return static_cast<H *>(b);
}

// Stage a host vector into workspace slot 0 behind kHeadroomBytes of headroom.
static int stage_in(const void *x_host, size_t bytes, void **x_dev)
{
    void *base = nullptr;
    int rc = ws_reserve(0, kHeadroomBytes + round_up(bytes, 256) + 256, &base);
    if (rc) return rc;
    *x_dev = (char *)base + kHeadroomBytes;
    if (bytes) SK_HIP(hipMemcpyAsync(*x_dev, x_host, bytes, hipMemcpyHostToDevice, ctx().stream));
    return SKDSP_OK;
}

static int stage_out(void *y_host, const void *y_dev, size_t bytes, const HandleBase *h = nullptr)
{
    if (bytes && h && h->wide_out && !dtype_double(h->dtype)) {
        // widen on the device (slot 0 held x, which the kernels are done with in stream order)
        void *wide = nullptr;
        int rc = ws_reserve(0, 2 * bytes + 256, &wide);
        if (rc) return rc;
        if ((rc = widen_launch(y_dev, (int64_t)(bytes / 4), wide, ctx().stream))) return rc;
        y_dev = wide;
        bytes *= 2;
    }
    if (bytes) SK_HIP(hipMemcpyAsync(y_host, y_dev, bytes, hipMemcpyDeviceToHost, ctx().stream));
    SK_HIP(hipStreamSynchronize(ctx().stream));
    return SKDSP_OK;
}

// IIR on an interleaved-or-real device vector (handles the complex -> 2 planes detour).
// tmp slot 3 holds the planes.  y may alias x.
static int iir_any_dev(IirHandle *h, const void *x_dev, int64_t n, void *y_dev, const double *zi = nullptr, double *zf = nullptr)
{
    hipStream_t s = ctx().stream;
    if (n <= 0) {
        const size_t zb = (size_t)(dtype_complex(h->dtype) ? 2 : 1) * h->nsec * h->order * 8;
        if (zf && zi) memcpy(zf, zi, zb);
        else if (zf) memset(zf, 0, zb);
        return SKDSP_OK;
    }
    if (!dtype_complex(h->dtype)) return iir_launch_planar(h, x_dev, n, 1, 0, y_dev, s, zi, zf);
    const bool planar_only = opt().iir_planar != 0;  // developer A/B switch (and the tests)
    if (!planar_only) {
        // decaying filters: both components stay interleaved end to end (iir_k1c / iir_k3c kernels)
        const int r1 = iir_launch_planar(h, x_dev, n, 2, 0, y_dev, s, zi, zf, 1);
        if (r1 != 1) return r1;
    }
    const size_t rsz = dtype_double(h->dtype) ? 8 : 4;
    const int64_t stride = (int64_t)round_up((size_t)n, 64);
    void *planes = nullptr;
    int rc = ws_reserve(3, (size_t)2 * stride * rsz, &planes);
    if (rc) return rc;
    void *re = planes, *im = (char *)planes + (size_t)stride * rsz;
    if ((rc = deinterleave_launch(x_dev, n, h->dtype, re, im, s))) return rc;
    if ((rc = iir_launch_planar(h, planes, n, 2, stride, planes, s, zi, zf))) return rc;
    return interleave_launch(re, im, n, h->dtype, y_dev, s);
}

// y = filter(L * upsample(x, L)).  Real: zero-stuff straight into y, then filter in place.  Complex:
// zero-stuff straight into the two planes the scan works on (no stuffed interleaved copy, no
// deinterleave pass), filter, interleave into y.
static int iir_up_any(IirHandle *h, const void *x_dev, int64_t n, int L, void *y_dev)
{
    hipStream_t s = ctx().stream;
    const int64_t nl = n * L;
    if (nl <= 0) return SKDSP_OK;
    int rc;
    if (!dtype_complex(h->dtype)) {
        if ((rc = upsample_launch(x_dev, n, L, h->dtype, (double)L, y_dev, s))) return rc;
        return iir_any_dev(h, y_dev, nl, y_dev);
    }
    const size_t rsz = dtype_double(h->dtype) ? 8 : 4;
    const int64_t stride = (int64_t)round_up((size_t)nl, 64);
    void *planes = nullptr;
    if ((rc = ws_reserve(3, (size_t)2 * stride * rsz, &planes))) return rc;
    void *re = planes, *im = (char *)planes + (size_t)stride * rsz;
    if ((rc = upsample_planes_launch(x_dev, n, L, h->dtype, (double)L, re, im, s))) return rc;
    if ((rc = iir_launch_planar(h, planes, nl, 2, stride, planes, s))) return rc;
    return interleave_launch(re, im, nl, h->dtype, y_dev, s);
}


// ---------------------------------------------------------------------------
// (b, a) -> cascaded biquads.  scipy.signal.lfilter runs a transfer function as ONE
// direct-form-II-transposed section of order N.  In those state coordinates the
// one-chunk transition matrix A^T of a narrow-band design (rate_change(12): Butterworth
// order 8, cutoff 0.075) has entries ~1e6 that cancel, so the affine scan would lose
// ~1e-4 of the output even in float64 (measured).  The scan therefore runs the SAME
// transfer function as second-order sections, whose state coordinates are benign; the
// result differs from the reference's TF-form recursion by its own float64 roundoff
// level (~1e-9 relative for rate_change(12), tests/golden/g8).  Conjugate pairs are
// symmetrised so every section has real coefficients.
typedef std::complex<long double> cld;

// Roots of c[0] z^n + ... + c[n] as the eigenvalues of the (real) companion matrix by
// the Francis double-shift QR iteration (the classical EISPACK "hqr" scheme) in long
// double.  Orthogonal similarity transforms are backward stable, and REAL arithmetic
// returns exactly conjugate pairs -- both matter for the N-fold zero at z = -1 of a
// Butterworth numerator: the individual roots scatter by eps^(1/N), yet the product of
// the resulting real quadratic factors reproduces the coefficients to ~1e-18 (an
// Aberth iteration, or a complex-shift QR followed by symmetrising the pairs, measured
// 1e-6 .. 1e-4 there).
static inline long double sign_ld(long double a, long double b) { return b >= 0.0L ? fabsl(a) : -fabsl(a); }

static bool poly_roots(const std::vector<long double> &c, std::vector<cld> &roots)
{
    const int n = (int)c.size() - 1;
    roots.clear();
    if (n <= 0) return true;
    std::vector<long double> A((size_t)n * n, 0.0L);
    auto a = [&](int i, int j) -> long double & { return A[(size_t)i * n + j]; };
    for (int j = 0; j < n; ++j) a(0, j) = -c[j + 1] / c[0];
    for (int i = 1; i < n; ++i) a(i, i - 1) = 1.0L;
    roots.assign(n, cld(0.0L, 0.0L));
    long double anorm = 0.0L;
    for (int i = 0; i < n; ++i)
        for (int j = (i > 0 ? i - 1 : 0); j < n; ++j) anorm += fabsl(a(i, j));
    int nn = n - 1;
    long double t = 0.0L, p = 0, q = 0, r = 0, s = 0, w = 0, x = 0, y = 0, z = 0;
    while (nn >= 0) {
        int its = 0, l;
        do {
            for (l = nn; l >= 1; --l) {
                s = fabsl(a(l - 1, l - 1)) + fabsl(a(l, l));
                if (s == 0.0L) s = anorm;
                if (fabsl(a(l, l - 1)) + s == s) { a(l, l - 1) = 0.0L; break; }
            }
            x = a(nn, nn);
            if (l == nn) {  // one root
                roots[nn--] = cld(x + t, 0.0L);
            } else {
                y = a(nn - 1, nn - 1);
                w = a(nn, nn - 1) * a(nn - 1, nn);
                if (l == nn - 1) {  // two roots
                    p = 0.5L * (y - x);
                    q = p * p + w;
                    z = sqrtl(fabsl(q));
                    x += t;
                    if (q >= 0.0L) {
                        z = p + sign_ld(z, p);
                        roots[nn - 1] = roots[nn] = cld(x + z, 0.0L);
                        if (z != 0.0L) roots[nn] = cld(x - w / z, 0.0L);
                    } else {
                        roots[nn - 1] = cld(x + p, z);
                        roots[nn] = cld(x + p, -z);
                    }
                    nn -= 2;
                } else {  // no roots yet: one double-shift sweep
                    if (its == 120) return false;
                    if (its % 10 == 0 && its > 0) {  // exceptional shift
                        t += x;
                        for (int i = 0; i <= nn; ++i) a(i, i) -= x;
                        s = fabsl(a(nn, nn - 1)) + fabsl(a(nn - 1, nn - 2));
                        y = x = 0.75L * s;
                        w = -0.4375L * s * s;
                    }
                    ++its;
                    int m;
                    for (m = nn - 2; m >= l; --m) {
                        z = a(m, m);
                        r = x - z;
                        s = y - z;
                        p = (r * s - w) / a(m + 1, m) + a(m, m + 1);
                        q = a(m + 1, m + 1) - z - r - s;
                        r = a(m + 2, m + 1);
                        s = fabsl(p) + fabsl(q) + fabsl(r);
                        p /= s; q /= s; r /= s;
                        if (m == l) break;
                        const long double u = fabsl(a(m, m - 1)) * (fabsl(q) + fabsl(r));
                        const long double v = fabsl(p) * (fabsl(a(m - 1, m - 1)) + fabsl(z) + fabsl(a(m + 1, m + 1)));
                        if (u + v == v) break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        a(i, i - 2) = 0.0L;
                        if (i != m + 2) a(i, i - 3) = 0.0L;
                    }
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = a(k, k - 1);
                            q = a(k + 1, k - 1);
                            r = 0.0L;
                            if (k != nn - 1) r = a(k + 2, k - 1);
                            if ((x = fabsl(p) + fabsl(q) + fabsl(r)) != 0.0L) { p /= x; q /= x; r /= x; }
                        }
                        if ((s = sign_ld(sqrtl(p * p + q * q + r * r), p)) != 0.0L) {
                            if (k == m) {
                                if (l != m) a(k, k - 1) = -a(k, k - 1);
                            } else {
                                a(k, k - 1) = -s * x;
                            }
                            p += s;
                            x = p / s; y = q / s; z = r / s;
                            q /= p; r /= p;
                            for (int j = k; j <= nn; ++j) {
                                p = a(k, j) + q * a(k + 1, j);
                                if (k != nn - 1) { p += r * a(k + 2, j); a(k + 2, j) -= p * z; }
                                a(k + 1, j) -= p * y;
                                a(k, j) -= p * x;
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * a(i, k) + y * a(i, k + 1);
                                if (k != nn - 1) { p += z * a(i, k + 2); a(i, k + 2) -= p * r; }
                                a(i, k + 1) -= p * q;
                                a(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    for (auto &rt : roots)
        if (!std::isfinite((double)rt.real()) || !std::isfinite((double)rt.imag())) return false;
    return true;
}

// group roots of a real polynomial into real quadratic factors 1 + c1 z^-1 + c2 z^-2
static bool quad_factors(std::vector<cld> roots, std::vector<std::pair<long double, long double>> &quads)
{
    quads.clear();
    std::vector<cld> up, dn;
    std::vector<long double> re;
    for (auto &r : roots) {
        const long double tol = 1e-13L * (1.0L + std::abs(r));
        if (r.imag() > tol) up.push_back(r);
        else if (r.imag() < -tol) dn.push_back(r);
        else re.push_back(r.real());
    }
    if (up.size() != dn.size()) return false;
    for (auto &u : up) {
        // nearest partner to conj(u)
        size_t best = 0;
        long double bd = -1.0L;
        for (size_t j = 0; j < dn.size(); ++j) {
            const long double d = std::abs(std::conj(u) - dn[j]);
            if (bd < 0.0L || d < bd) { bd = d; best = j; }
        }
        const cld z = u;  // hqr returns exact conjugate pairs
        dn.erase(dn.begin() + (long)best);
        quads.push_back({-2.0L * z.real(), std::norm(z)});
    }
    std::sort(re.begin(), re.end());
    for (size_t i = 0; i + 1 < re.size(); i += 2) quads.push_back({-(re[i] + re[i + 1]), re[i] * re[i + 1]});
    if (re.size() & 1) quads.push_back({-re.back(), 0.0L});
    return true;
}

static int tf_to_sos(const double *b, int nb, const double *a, int na, std::vector<double> &sos, int *nsec_out)
{
    // normalise by a[0]; strip trailing zeros (roots at the origin contribute a unit factor)
    std::vector<long double> bb(b, b + nb), aa(a, a + na);
    for (auto &v : bb) v /= (long double)a[0];
    for (auto &v : aa) v /= (long double)a[0];
    while (bb.size() > 1 && bb.back() == 0.0L) bb.pop_back();
    while (aa.size() > 1 && aa.back() == 0.0L) aa.pop_back();
    int delay = 0;  // leading zeros of b = pure delays z^-delay
    while (bb.size() > 1 && bb.front() == 0.0L) { bb.erase(bb.begin()); ++delay; }
    const long double gain = bb.front();
    std::vector<std::pair<long double, long double>> zq, pq;
    if (gain != 0.0L) {
        std::vector<cld> zr;
        SK_CHECK(poly_roots(bb, zr) && quad_factors(zr, zq), SKDSP_ERR_UNSUPPORTED,
                 "tf_create: could not factor the numerator into real second-order sections");
    }
    std::vector<cld> pr;
    SK_CHECK(poly_roots(aa, pr) && quad_factors(pr, pq), SKDSP_ERR_UNSUPPORTED,
             "tf_create: could not factor the denominator into real second-order sections");
    // delays become numerator factors z^-1 / z^-2
    std::vector<std::array<long double, 3>> num;
    for (auto &q : zq) num.push_back({1.0L, q.first, q.second});
    for (; delay >= 2; delay -= 2) num.push_back({0.0L, 0.0L, 1.0L});
    if (delay == 1) num.push_back({0.0L, 1.0L, 0.0L});
    const size_t ns = std::max<size_t>(std::max(num.size(), pq.size()), 1);
    SK_CHECK(ns <= 12, SKDSP_ERR_UNSUPPORTED, "tf_create: order %d needs more than 12 second-order sections",
             (int)std::max(nb, na) - 1);
    // sections in order of increasing pole radius (quiet sections first), gain on the first
    std::sort(pq.begin(), pq.end(), [](const auto &x, const auto &y) { return x.second < y.second; });
    sos.assign(ns * 6, 0.0);
    for (size_t s = 0; s < ns; ++s) {
        std::array<long double, 3> nmr = s < num.size() ? num[s] : std::array<long double, 3>{1.0L, 0.0L, 0.0L};
        if (s == 0) for (auto &v : nmr) v *= gain;
        sos[6 * s + 0] = (double)nmr[0];
        sos[6 * s + 1] = (double)nmr[1];
        sos[6 * s + 2] = (double)nmr[2];
        sos[6 * s + 3] = 1.0;
        sos[6 * s + 4] = s < pq.size() ? (double)pq[s].first : 0.0;
        sos[6 * s + 5] = s < pq.size() ? (double)pq[s].second : 0.0;
    }
    *nsec_out = (int)ns;
    return SKDSP_OK;
}

}  // namespace skdsp

using namespace skdsp;

#define API_BEGIN                        \
    {                                    \
        int _rc = ensure_init();         \
        if (_rc) return _rc;             \
    }                                    \
    std::lock_guard<std::mutex> _ctxlk(ctx().mu)

extern "C" {

const char *skdsp_last_error(void) { return g_err; }
const char *skdsp_version(void) { return "skdsp-hip 0.1.0 (gfx950)"; }

int skdsp_init(int device)
{
    std::lock_guard<std::mutex> lk(ctx().mu);
    return init_locked(device);
}

int skdsp_shutdown(void)
{
    Context &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.ready) return SKDSP_OK;
    (void)hipStreamSynchronize(c.stream);
    for (int i = 0; i < 4; ++i) {
        if (c.ws[i]) (void)hipFree(c.ws[i]);
        c.ws[i] = nullptr;
        c.ws_bytes[i] = 0;
    }
    (void)hipEventDestroy(c.ev_start);
    (void)hipEventDestroy(c.ev_stop);
    if (c.comm_stream) {
        (void)hipStreamSynchronize(c.comm_stream);
        (void)hipEventDestroy(c.ev_in);
        (void)hipEventDestroy(c.ev_halo);
        if (c.halo_flag) (void)hipFree(c.halo_flag);
        if (c.halo_err) (void)hipHostFree(c.halo_err);
        c.halo_flag = nullptr; c.halo_err = nullptr;
        (void)hipStreamDestroy(c.comm_stream);
        c.comm_stream = nullptr;
        c.ev_in = c.ev_halo = nullptr;
    }
    (void)hipStreamDestroy(c.stream);
    c.ready = false;
    c.device = -1;
    return SKDSP_OK;
}

int skdsp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int skdsp_device_info(char *name, int name_cap, int *compute_units, int64_t *hbm_bytes, int *clock_khz)
{
    API_BEGIN;
    hipDeviceProp_t prop;
    SK_HIP(hipGetDeviceProperties(&prop, ctx().device));
    if (name && name_cap > 0) {
        snprintf(name, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (clock_khz) *clock_khz = prop.clockRate;
    return SKDSP_OK;
}

int skdsp_malloc(void **dptr, int64_t bytes)
{
    API_BEGIN;
    SK_CHECK(dptr && bytes >= 0, SKDSP_ERR_BADARG, "skdsp_malloc: bad arguments");
    SK_HIP(hipMalloc(dptr, (size_t)(bytes > 0 ? bytes : 1)));
    return SKDSP_OK;
}
int skdsp_free(void *dptr)
{
    API_BEGIN;
    if (dptr) {
        SK_HIP(hipStreamSynchronize(ctx().stream));
        SK_HIP(hipFree(dptr));
    }
    return SKDSP_OK;
}
int skdsp_memcpy_h2d(void *dst, const void *src, int64_t bytes)
{
    API_BEGIN;
    if (bytes > 0) SK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx().stream));
    SK_HIP(hipStreamSynchronize(ctx().stream));
    return SKDSP_OK;
}
int skdsp_memcpy_d2h(void *dst, const void *src, int64_t bytes)
{
    API_BEGIN;
    if (bytes > 0) SK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx().stream));
    SK_HIP(hipStreamSynchronize(ctx().stream));
    return SKDSP_OK;
}
int skdsp_memcpy_d2d(void *dst, const void *src, int64_t bytes)
{
    API_BEGIN;
    if (bytes > 0) SK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, ctx().stream));
    return SKDSP_OK;
}
int skdsp_memset(void *dst, int value, int64_t bytes)
{
    API_BEGIN;
    if (bytes > 0) SK_HIP(hipMemsetAsync(dst, value, (size_t)bytes, ctx().stream));
    return SKDSP_OK;
}
int skdsp_sync(void)
{
    API_BEGIN;
    SK_HIP(hipStreamSynchronize(ctx().stream));
    return SKDSP_OK;
}
int skdsp_timer_start(void)
{
    API_BEGIN;
    SK_HIP(hipEventRecord(ctx().ev_start, ctx().stream));
    return SKDSP_OK;
}
int skdsp_timer_stop(float *ms)
{
    API_BEGIN;
    SK_HIP(hipEventRecord(ctx().ev_stop, ctx().stream));
    SK_HIP(hipEventSynchronize(ctx().ev_stop));
    float t = 0.f;
    SK_HIP(hipEventElapsedTime(&t, ctx().ev_start, ctx().ev_stop));
    if (ms) *ms = t;
    return SKDSP_OK;
}
int skdsp_fill_noise_dev(void *x_dev, int64_t n, int dtype, uint64_t seed, int64_t first_index)
{
    API_BEGIN;
    SK_CHECK(dtype_valid(dtype), SKDSP_ERR_BADARG, "fill_noise: bad dtype %d", dtype);
    return fill_noise_launch(x_dev, n, dtype, seed, first_index, ctx().stream);
}

// ------------------------------------------------------------------------ FIR
int skdsp_fir_create(const void *taps, int ntaps, int taps_complex, int dtype, skdsp_handle *out)
{
    API_BEGIN;
    SK_CHECK(out, SKDSP_ERR_BADARG, "fir_create: null out");
    SK_CHECK(taps && ntaps >= 1, SKDSP_ERR_BADARG, "fir_create: need at least one tap");
    SK_CHECK(dtype_valid(dtype), SKDSP_ERR_BADARG, "fir_create: bad dtype %d", dtype);
    SK_CHECK(!(taps_complex && !dtype_complex(dtype)), SKDSP_ERR_BADARG,
             "fir_create: complex taps need a complex signal dtype");
    std::unique_ptr<FirHandle> h(new FirHandle());
    h->kind = H_FIR;
    h->dtype = dtype;
    h->ntaps = ntaps;
    h->taps_complex = taps_complex != 0;
    const int comp = taps_complex ? 2 : 1;
    h->taps_host.assign((const double *)taps, (const double *)taps + (size_t)ntaps * comp);
    *out = h.release();
    return SKDSP_OK;
}

int skdsp_fir_set_algo(skdsp_handle hh, int algo)
{
    FirHandle *h = as_handle<FirHandle>(hh, H_FIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "fir_set_algo: not a FIR handle");
    SK_CHECK(algo >= SKDSP_FIR_AUTO && algo <= SKDSP_FIR_OLS, SKDSP_ERR_BADARG, "fir_set_algo: bad algo %d", algo);
    h->algo = algo;
    return SKDSP_OK;
}

int skdsp_fir_get_algo(skdsp_handle hh, int64_t n, int *algo_used)
{
    FirHandle *h = as_handle<FirHandle>(hh, H_FIR);
    SK_CHECK(h && algo_used, SKDSP_ERR_BADARG, "fir_get_algo: bad arguments");
    *algo_used = pick_fir_algo(h, n);
    return SKDSP_OK;
}

int skdsp_fir_filter_dev(skdsp_handle hh, const void *x_dev, int64_t n, int64_t n_hist, void *y_dev)
{
    API_BEGIN;
    FirHandle *h = as_handle<FirHandle>(hh, H_FIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "fir_filter: not a FIR handle");
    SK_CHECK(n >= 0 && n_hist >= 0, SKDSP_ERR_BADARG, "fir_filter: negative length");
    std::lock_guard<std::mutex> lk(h->mu);
    return fir_filter_any(h, x_dev, n, n_hist, y_dev);
}

int skdsp_fir_up_dev(skdsp_handle hh, const void *x_dev, int64_t n, int64_t n_hist, int L, void *y_dev)
{
    API_BEGIN;
    FirHandle *h = as_handle<FirHandle>(hh, H_FIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "fir_up: not a FIR handle");
    SK_CHECK(L >= 1, SKDSP_ERR_BADARG, "fir_up: L must be >= 1");
    std::lock_guard<std::mutex> lk(h->mu);
    return fir_direct_launch(h, x_dev, n, n_hist, L, 1, n * L, y_dev, ctx().stream);
}

int skdsp_fir_dn_dev(skdsp_handle hh, const void *x_dev, int64_t n, int64_t n_hist, int M, void *y_dev)
{
    API_BEGIN;
    FirHandle *h = as_handle<FirHandle>(hh, H_FIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "fir_dn: not a FIR handle");
    SK_CHECK(M >= 1, SKDSP_ERR_BADARG, "fir_dn: M must be >= 1");
    std::lock_guard<std::mutex> lk(h->mu);
    return fir_dn_any(h, x_dev, n, n_hist, M, y_dev);
}

int skdsp_fir_updn_dev(skdsp_handle hh, const void *x_dev, int64_t n, int64_t n_hist, int L, int M, void *y_dev)
{
    API_BEGIN;
    FirHandle *h = as_handle<FirHandle>(hh, H_FIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "fir_updn: not a FIR handle");
    SK_CHECK(L >= 1 && M >= 1, SKDSP_ERR_BADARG, "fir_updn: L, M must be >= 1");
    std::lock_guard<std::mutex> lk(h->mu);
    return fir_direct_launch(h, x_dev, n, n_hist, L, M, (n * L) / M, y_dev, ctx().stream);
}

static int fir_host_call(skdsp_handle hh, const void *x, int64_t n, int L, int M, int mode, void *y)
{
    API_BEGIN;
    FirHandle *h = as_handle<FirHandle>(hh, H_FIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "fir: not a FIR handle");
    SK_CHECK(n >= 0 && L >= 1 && M >= 1, SKDSP_ERR_BADARG, "fir: bad arguments (n=%lld L=%d M=%d)", (long long)n, L, M);
    const size_t esz = dtype_size(h->dtype);
    const int64_t n_out = mode == 0 ? n : (n * L) / M;
    if (n_out == 0) return SKDSP_OK;
    SK_CHECK(x && y, SKDSP_ERR_BADARG, "fir: null buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    void *x_dev = nullptr, *y_dev = nullptr;
    int rc = stage_in(x, (size_t)n * esz, &x_dev);
    if (rc) return rc;
    if ((rc = ws_reserve(1, (size_t)n_out * esz + 256, &y_dev))) return rc;
    if (mode == 0) rc = fir_filter_any(h, x_dev, n, 0, y_dev);
    else if (L == 1) rc = fir_dn_any(h, x_dev, n, 0, M, y_dev);
    else rc = fir_direct_launch(h, x_dev, n, 0, L, M, n_out, y_dev, ctx().stream);
    if (rc) return rc;
    return stage_out(y, y_dev, (size_t)n_out * esz, h);
}

int skdsp_fir_filter(skdsp_handle h, const void *x, int64_t n, void *y) { return fir_host_call(h, x, n, 1, 1, 0, y); }
int skdsp_fir_up(skdsp_handle h, const void *x, int64_t n, int L, void *y) { return fir_host_call(h, x, n, L, 1, 1, y); }
int skdsp_fir_dn(skdsp_handle h, const void *x, int64_t n, int M, void *y) { return fir_host_call(h, x, n, 1, M, 1, y); }
int skdsp_fir_updn(skdsp_handle h, const void *x, int64_t n, int L, int M, void *y) { return fir_host_call(h, x, n, L, M, 1, y); }

// ------------------------------------------------------------------------ IIR
static int iir_create_common(int nsec, int order, const std::vector<double> &coef, int dtype, skdsp_handle *out)
{
    SK_CHECK(out, SKDSP_ERR_BADARG, "iir_create: null out");
    SK_CHECK(dtype_valid(dtype), SKDSP_ERR_BADARG, "iir_create: bad dtype %d", dtype);
    SK_CHECK(iir_shape_supported(nsec, order), SKDSP_ERR_UNSUPPORTED,
             "iir_create: %d sections of order %d not supported (SOS: 1..12 sections; (b,a): order 1..12)", nsec, order);
    std::unique_ptr<IirHandle> h(new IirHandle());
    h->kind = H_IIR;
    h->dtype = dtype;
    h->nsec = nsec;
    h->order = order;
    h->coef = coef;
    if (order == 2 && nsec >= 2 && !opt().iir_no_unit) {
        // unit-tail re-factorisation (see IirHandle): H_0' = H_0 * prod_{j>=1} b0_j,  H_k' = H_k / b0_k
        bool ok = true;
        for (int s = 0; s < nsec && ok; ++s) {
            const double *c = coef.data() + 5 * s;
            ok = c[0] != 0.0 && std::isfinite(c[0]) && (s == 0 || std::fabs(c[2] / c[0] - 1.0) <= 1e-13);
        }
        if (ok) {
            std::vector<long double> tail((size_t)nsec + 1, 1.0L);  // tail[k] = prod_{j>=k} b0_j
            for (int s = nsec - 1; s >= 0; --s) tail[s] = tail[s + 1] * (long double)coef[5 * s];
            for (int s = 0; s < nsec && ok; ++s) ok = std::isfinite((double)tail[s]) && tail[s] != 0.0L;
            if (ok) {
                h->state_scale.resize((size_t)2 * nsec);
                for (int s = 0; s < nsec; ++s) {
                    double *c = h->coef.data() + 5 * s;
                    if (s == 0) {
                        for (int k = 0; k < 3; ++k) c[k] = (double)((long double)c[k] * tail[1]);
                    } else {
                        const long double b0 = c[0];
                        c[1] = (double)((long double)c[1] / b0);
                        c[0] = 1.0;
                        c[2] = 1.0;
                    }
                    h->state_scale[2 * s] = h->state_scale[2 * s + 1] = (double)tail[s + 1];
                }
                h->unit_tail = true;
            }
        }
    }
    *out = h.release();
    return SKDSP_OK;
}

int skdsp_sos_create(const double *sos, int nsec, int dtype, skdsp_handle *out)
{
    API_BEGIN;
    SK_CHECK(sos && nsec >= 1, SKDSP_ERR_BADARG, "sos_create: sos array must be shape (n_sections, 6)");
    std::vector<double> coef((size_t)nsec * 5);
    for (int s = 0; s < nsec; ++s) {
        const double *q = sos + 6 * s;
        SK_CHECK(q[3] == 1.0, SKDSP_ERR_BADARG, "sos[:, 3] should be all ones");
        double *c = coef.data() + 5 * s;
        c[0] = q[0]; c[1] = q[1]; c[2] = q[2]; c[3] = q[4]; c[4] = q[5];
    }
    return iir_create_common(nsec, 2, coef, dtype, out);
}

int skdsp_tf2sos(const double *b, int nb, const double *a, int na, double *sos_out, int *nsec_out)
{
    // host-only helper (no GPU needed): the factorisation skdsp_tf_create applies
    SK_CHECK(b && a && nb >= 1 && na >= 1 && sos_out && nsec_out, SKDSP_ERR_BADARG, "tf2sos: bad arguments");
    SK_CHECK(a[0] != 0.0, SKDSP_ERR_BADARG, "tf2sos: a[0] must be nonzero");
    std::vector<double> sos;
    int nsec = 0;
    int rc = tf_to_sos(b, nb, a, na, sos, &nsec);
    if (rc) return rc;
    memcpy(sos_out, sos.data(), sos.size() * sizeof(double));
    *nsec_out = nsec;
    return SKDSP_OK;
}

int skdsp_tf_create(const double *b, int nb, const double *a, int na, int dtype, skdsp_handle *out)
{
    API_BEGIN;
    SK_CHECK(b && a && nb >= 1 && na >= 1, SKDSP_ERR_BADARG, "tf_create: need b and a");
    SK_CHECK(a[0] != 0.0, SKDSP_ERR_BADARG, "tf_create: a[0] must be nonzero");
    std::vector<double> sos;
    int nsec = 0;
    int rc = tf_to_sos(b, nb, a, na, sos, &nsec);
    if (rc) return rc;
    std::vector<double> coef((size_t)nsec * 5);
    for (int s = 0; s < nsec; ++s) {
        const double *q = sos.data() + 6 * s;
        double *c = coef.data() + 5 * s;
        c[0] = q[0]; c[1] = q[1]; c[2] = q[2]; c[3] = q[4]; c[4] = q[5];
    }
    return iir_create_common(nsec, 2, coef, dtype, out);
}

int skdsp_iir_filter_dev(skdsp_handle hh, const void *x_dev, int64_t n, void *y_dev)
{
    API_BEGIN;
    IirHandle *h = as_handle<IirHandle>(hh, H_IIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "iir_filter: not an IIR handle");
    std::lock_guard<std::mutex> lk(h->mu);
    return iir_any_dev(h, x_dev, n, y_dev);
}

int skdsp_iir_state_len(skdsp_handle hh, int *len)
{
    IirHandle *h = as_handle<IirHandle>(hh, H_IIR);
    SK_CHECK(h && len, SKDSP_ERR_BADARG, "iir_state_len: not an IIR handle");
    *len = (dtype_complex(h->dtype) ? 2 : 1) * h->nsec * h->order;
    return SKDSP_OK;
}

int skdsp_iir_filter_state_dev(skdsp_handle hh, const void *x_dev, int64_t n, const double *zi, double *zf, void *y_dev)
{
    API_BEGIN;
    IirHandle *h = as_handle<IirHandle>(hh, H_IIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "iir_filter_state: not an IIR handle");
    std::lock_guard<std::mutex> lk(h->mu);
    return iir_any_dev(h, x_dev, n, y_dev, zi, zf);
}

int skdsp_iir_up_dev(skdsp_handle hh, const void *x_dev, int64_t n, int L, void *y_dev)
{
    API_BEGIN;
    IirHandle *h = as_handle<IirHandle>(hh, H_IIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "iir_up: not an IIR handle");
    SK_CHECK(L >= 1, SKDSP_ERR_BADARG, "iir_up: L must be >= 1");
    std::lock_guard<std::mutex> lk(h->mu);
    return iir_up_any(h, x_dev, n, L, y_dev);
}

// y = downsample(filter(x), M) on device vectors (both the _dev and the host-pointer entry use it)
static int iir_dn_any(IirHandle *h, const void *x_dev, int64_t n, int M, void *y_dev)
{
    if (n <= 0) return SKDSP_OK;
    // K3 stores every M-th output itself -- the full-rate result never reaches HBM
    if (M > 1 && M <= 4096 && !opt().iir_dn_full) {
        if (!dtype_complex(h->dtype)) return iir_launch_planar(h, x_dev, n, 1, 0, y_dev, ctx().stream, nullptr, nullptr, 0, M);
        if (!opt().iir_planar) {  // interleaved complex kernels (decaying filters); 1 = not applicable
            const int r1 = iir_launch_planar(h, x_dev, n, 2, 0, y_dev, ctx().stream, nullptr, nullptr, 1, M);
            if (r1 != 1) return r1;
        }
    }
    void *full = nullptr;
    int rc = ws_reserve(2, (size_t)n * dtype_size(h->dtype) + 256, &full);
    if (rc) return rc;
    if ((rc = iir_any_dev(h, x_dev, n, full))) return rc;
    return downsample_launch(full, n, M, 0, h->dtype, y_dev, ctx().stream);
}

int skdsp_iir_dn_dev(skdsp_handle hh, const void *x_dev, int64_t n, int M, void *y_dev)
{
    API_BEGIN;
    IirHandle *h = as_handle<IirHandle>(hh, H_IIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "iir_dn: not an IIR handle");
    SK_CHECK(M >= 1, SKDSP_ERR_BADARG, "iir_dn: M must be >= 1");
    std::lock_guard<std::mutex> lk(h->mu);
    return iir_dn_any(h, x_dev, n, M, y_dev);
}

static int iir_host_call(skdsp_handle hh, const void *x, int64_t n, int L, int M, void *y)
{
    API_BEGIN;
    IirHandle *h = as_handle<IirHandle>(hh, H_IIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "iir: not an IIR handle");
    SK_CHECK(n >= 0 && L >= 1 && M >= 1, SKDSP_ERR_BADARG, "iir: bad arguments");
    const size_t esz = dtype_size(h->dtype);
    const int64_t n_out = (n * L) / M;
    if (n_out == 0) return SKDSP_OK;  // fewer than M samples: nothing to deliver (y may be NULL)
    SK_CHECK(x && y, SKDSP_ERR_BADARG, "iir: null buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    void *x_dev = nullptr, *y_dev = nullptr;
    int rc = stage_in(x, (size_t)n * esz, &x_dev);
    if (rc) return rc;
    if (L > 1) {
        if ((rc = ws_reserve(1, (size_t)n * L * esz + 256, &y_dev))) return rc;
        if ((rc = iir_up_any(h, x_dev, n, L, y_dev))) return rc;
    } else if (M > 1) {
        if ((rc = ws_reserve(1, (size_t)n_out * esz + 256, &y_dev))) return rc;
        if ((rc = iir_dn_any(h, x_dev, n, M, y_dev))) return rc;  // K3 stores every M-th output itself
    } else {
        if ((rc = ws_reserve(1, (size_t)n * esz + 256, &y_dev))) return rc;
        if ((rc = iir_any_dev(h, x_dev, n, y_dev))) return rc;
    }
    return stage_out(y, y_dev, (size_t)n_out * esz, h);
}

int skdsp_iir_filter(skdsp_handle h, const void *x, int64_t n, void *y) { return iir_host_call(h, x, n, 1, 1, y); }
int skdsp_iir_up(skdsp_handle h, const void *x, int64_t n, int L, void *y) { return iir_host_call(h, x, n, L, 1, y); }
int skdsp_iir_dn(skdsp_handle h, const void *x, int64_t n, int M, void *y) { return iir_host_call(h, x, n, 1, M, y); }

// ---------------------------------------------------------------- resamplers
int skdsp_upsample_dev(const void *x_dev, int64_t n, int L, int dtype, double scale, void *y_dev)
{
    API_BEGIN;
    SK_CHECK(dtype_valid(dtype), SKDSP_ERR_BADARG, "upsample: bad dtype %d", dtype);
    return upsample_launch(x_dev, n, L, dtype, scale, y_dev, ctx().stream);
}

int skdsp_downsample_dev(const void *x_dev, int64_t n, int M, int p, int dtype, void *y_dev)
{
    API_BEGIN;
    SK_CHECK(dtype_valid(dtype), SKDSP_ERR_BADARG, "downsample: bad dtype %d", dtype);
    return downsample_launch(x_dev, n, M, p, dtype, y_dev, ctx().stream);
}

int skdsp_upsample(const void *x, int64_t n, int L, int dtype, void *y)
{
    API_BEGIN;
    SK_CHECK(dtype_valid(dtype) && n >= 0 && L >= 1, SKDSP_ERR_BADARG, "upsample: bad arguments");
    if (n == 0) return SKDSP_OK;
    const size_t esz = dtype_size(dtype);
    void *x_dev = nullptr, *y_dev = nullptr;
    int rc = stage_in(x, (size_t)n * esz, &x_dev);
    if (rc) return rc;
    if ((rc = ws_reserve(1, (size_t)n * L * esz + 256, &y_dev))) return rc;
    if ((rc = upsample_launch(x_dev, n, L, dtype, 1.0, y_dev, ctx().stream))) return rc;
    return stage_out(y, y_dev, (size_t)n * L * esz);
}

int skdsp_downsample(const void *x, int64_t n, int M, int p, int dtype, void *y)
{
    API_BEGIN;
    SK_CHECK(dtype_valid(dtype) && n >= 0 && M >= 1, SKDSP_ERR_BADARG, "downsample: bad arguments");
    SK_CHECK(p >= 0 && p < M, SKDSP_ERR_BADARG, "downsample: phase p=%d out of range for M=%d", p, M);
    const int64_t n_out = n / M;
    if (n_out == 0) return SKDSP_OK;
    const size_t esz = dtype_size(dtype);
    void *x_dev = nullptr, *y_dev = nullptr;
    int rc = stage_in(x, (size_t)n * esz, &x_dev);
    if (rc) return rc;
    if ((rc = ws_reserve(1, (size_t)n_out * esz + 256, &y_dev))) return rc;
    if ((rc = downsample_launch(x_dev, n, M, p, dtype, y_dev, ctx().stream))) return rc;
    return stage_out(y, y_dev, (size_t)n_out * esz);
}

int skdsp_set_option(const char *name, int value)
{
    SK_CHECK(name, SKDSP_ERR_BADARG, "set_option: null name");
    for (const OptEntry &e : kOptTable)
        if (!strcmp(e.name, name)) {
            opt().*(e.field) = value;
            return SKDSP_OK;
        }
    SK_CHECK(false, SKDSP_ERR_BADARG, "set_option: unknown option '%s'", name);
}

int skdsp_get_option(const char *name, int *value)
{
    SK_CHECK(name && value, SKDSP_ERR_BADARG, "get_option: null argument");
    for (const OptEntry &e : kOptTable)
        if (!strcmp(e.name, name)) {
            *value = opt().*(e.field);
            return SKDSP_OK;
        }
    SK_CHECK(false, SKDSP_ERR_BADARG, "get_option: unknown option '%s'", name);
}

int skdsp_set_wide_output(skdsp_handle hh, int on)
{
    HandleBase *b = reinterpret_cast<HandleBase *>(hh);
    SK_CHECK(b && (b->kind == H_FIR || b->kind == H_IIR), SKDSP_ERR_BADARG, "set_wide_output: not a filter handle");
    std::lock_guard<std::mutex> lk(b->mu);
    b->wide_out = on != 0;
    return SKDSP_OK;
}

int skdsp_destroy(skdsp_handle hh)
{
    if (!hh) return SKDSP_OK;
    HandleBase *b = reinterpret_cast<HandleBase *>(hh);
    if (ctx().ready) {
        std::lock_guard<std::mutex> lk(ctx().mu);
        (void)hipStreamSynchronize(ctx().stream);
        delete b;
    } else {
        delete b;
    }
    return SKDSP_OK;
}

}  // extern "C"
