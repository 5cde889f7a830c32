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
#include <numeric>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <functional>

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

static Context g_slots[kMaxSlots];
static int g_nslots = 0;            // bound slots; slot 0 is bound by the first call that needs a device
static std::mutex g_slots_mu;
static thread_local int t_slot = 0;

Context &ctx() { return g_slots[t_slot]; }
Context &ctx_of(int slot) { return g_slots[slot]; }
int slot_count() { return g_nslots; }
int select_slot(int slot)
{
    SK_CHECK(slot >= 0 && slot < kMaxSlots && g_slots[slot].ready, SKDSP_ERR_BADARG, "select_slot: slot %d is not bound", slot);
    t_slot = slot;
    SK_HIP(hipSetDevice(g_slots[slot].device));
    return SKDSP_OK;
}

// ---- failures reported after the fact (bounded device-side waits) ------------------------------
unsigned *async_err_dev(int which)
{
    Context &c = ctx();
    if (!c.async_err) {
        if (hipHostMalloc((void **)&c.async_err, kAsyncErrWords * sizeof(unsigned), hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            c.async_err = nullptr;
            return nullptr;
        }
        for (int i = 0; i < kAsyncErrWords; ++i) c.async_err[i] = 0;
    }
    unsigned *dev = nullptr;
    if (hipHostGetDevicePointer((void **)&dev, c.async_err, 0) != hipSuccess) return nullptr;
    return dev + which;
}

int async_err_check(Context &c)
{
    if (!c.async_err) return SKDSP_OK;
    volatile unsigned *w = c.async_err;
    if (w[kAsyncErrHalo]) {
        // The persistent launch of a sharded FIR step polled for seconds and gave up: the RCCL receive did not run beside
        // it on this system.  Tile 0 of that step was not written from a valid halo.  The two-launch form is used from now
        // on; the caller repeats the step -- COLLECTIVELY, on every rank (each step is one send/recv pair).
        w[kAsyncErrHalo] = 0;
        opt().shard_two_launches = 1;
        set_error("fir_filter_shard: a sharded step since the last synchronisation gave up waiting for its halo inside the filter "
                  "launch (the first tile of that step is invalid); switched to the two-launch form (option shard_two_launches) -- "
                  "repeat the step on every rank");
        return SKDSP_ERR_RCCL;
    }
    if (w[kAsyncErrIirLookback]) {
        w[kAsyncErrIirLookback] = 0;
        set_error("iir: a look-back poll of a single-pass scan launched since the last synchronisation timed out (the results of "
                  "that call are invalid; option iir_two_pass = 1 selects the two-pass scan)");
        return SKDSP_ERR_HIP;
    }
    return SKDSP_OK;
}

// stream sync of the calling slot + the deferred failures of what ran on it
static int sync_checked()
{
    SK_HIP(hipStreamSynchronize(ctx().stream));
    return async_err_check(ctx());
}

// ---- options: environment read once, skdsp_set_option afterwards ----------------------------
namespace {
struct OptEntry { const char *name; int Options::*field; };
const OptEntry kOptTable[] = {
    {"device", &Options::device}, {"fir_algo", &Options::fir_algo}, {"dn_no_ols", &Options::dn_no_ols},
    {"fir_mm", &Options::fir_mm}, {"fir_bx", &Options::fir_bx}, 
    
    {"ols_reserve", &Options::ols_reserve}, {"fir_bx_t16", &Options::fir_bx_t16}, {"ols_keep_overlap", &Options::ols_keep_overlap}, {"fir_dn_fold", &Options::fir_dn_fold}, {"fir_up_rep", &Options::fir_up_rep}, {"iir_seq", &Options::iir_seq}, {"iir_up_jump", &Options::iir_up_jump}, {"iir_dn_t96", &Options::iir_dn_t96}, {"iir_up_lean", &Options::iir_up_lean}, {"iir_planar", &Options::iir_planar}, 
    {"iir_dn_full", &Options::iir_dn_full}, {"iir_no_mfma", &Options::iir_no_mfma}, 
    {"iir_two_pass", &Options::iir_two_pass}, {"iir_par", &Options::iir_par}, {"iir_par_v32", &Options::iir_par_v32}, {"iir_up_fused", &Options::iir_up_fused}, {"fir_up_ols_min", &Options::fir_up_ols_min}, {"fir_updn_fused", &Options::fir_updn_fused}, {"fir_up4k", &Options::fir_up4k}, {"fir_up4k_group", &Options::fir_up4k_group}, {"fir_up4k_staged", &Options::fir_up4k_staged}, {"fir_up2k", &Options::fir_up2k}, {"fir_dn4k", &Options::fir_dn4k}, {"fir_up_pair", &Options::fir_up_pair}, {"fir_up_rows_min", &Options::fir_up_rows_min}, {"iir_dn_compact", &Options::iir_dn_compact}, 
    {"shard_two_launches", &Options::shard_two_launches}, {"shard_probe", &Options::shard_probe}, {"shard_halo_state", &Options::shard_halo_state},
    {"shard_self_halo", &Options::shard_self_halo}, {"dist_force_comm", &Options::dist_force_comm},
    {"host_chunk_log2", &Options::host_chunk_log2}, {"host_pipeline", &Options::host_pipeline}, {"host_multi_slot", &Options::host_multi_slot},
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
                 "skdsp_init: slot %d is already bound to device %d", c.slot, c.device);
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
    c.slot = t_slot;
    c.ready = true;
    {
        std::lock_guard<std::mutex> lk(g_slots_mu);
        if (g_nslots < t_slot + 1) g_nslots = t_slot + 1;
    }
    return SKDSP_OK;
}

int ensure_init()
{
    Context &c = ctx();
    if (c.ready) {
        // several slots: make sure this thread's HIP device is the slot's (threads start on device 0)
        if (g_nslots > 1) SK_HIP(hipSetDevice(c.device));
        return SKDSP_OK;
    }
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
    const bool ols64 = fir_ols64_supported(h);
    if (algo == SKDSP_FIR_OLS && !fir_ols_supported(h) && !ols64) algo = SKDSP_FIR_DIRECT;
    if (algo != SKDSP_FIR_AUTO) return algo;
    // float64 signals: the direct form costs 2 (4 for complex taps) FP64 FMA per tap and real sample; the float64
    // overlap-save tile is flat in the tap count (measured crossovers at 2^26 samples: see DESIGN.md 4.2, LABNOTES.md)
    if (ols64) return h->ntaps >= (h->dtype == SKDSP_C128 ? 24 : 128) && n >= 8192 ? SKDSP_FIR_OLS : SKDSP_FIR_DIRECT;
    // measured crossover at 2^26 samples (tools/time_fir_filter.py, profiles/r04/fir_filter.txt): the matrix-pipe kernel (real taps, fp16 pieces)
    // stays ahead of overlap-save up to 6 lag blocks for complex64 (0.215 vs 0.229 ms at 145 taps; 0.221 vs 0.227 at 160; 0.241 vs 0.227 at 192)
    // and for float32 (0.109 vs 0.125 ms at 145 taps; 0.125 vs 0.126 at 192; 0.133 vs 0.122 at 224)
    const int ols_from = h->taps_complex ? 48 : (h->dtype == SKDSP_C64 ? 177 : 193);
    if (fir_ols_supported(h) && h->ntaps >= ols_from && n >= 4096) return SKDSP_FIR_OLS;
    return SKDSP_FIR_DIRECT;
}

int fir_algo_for(const FirHandle *h, int64_t n) { return pick_fir_algo(h, n); }

// .dn: long filters with a modest M go through the overlap-save engine with a decimating store, which
// beats Ntaps/M direct taps per kept sample (2^24 complex64, 512 taps, M = 3: 0.163 -> 0.085 ms).  Where the
// matrix-pipe kernel covers the geometry it is the faster one (profiles/r04/fir_dn.txt) except for the long filters of M <= 4: fir_dn_any.
// ---- tap partitioning: filters longer than one launch takes ------------------------------------------------------
// The reference accepts any tap count (lfilter(b,[1],x), multirate_helper.py:108).  One launch takes up to 4097 taps in
// the overlap-save engine (float32 / complex64) and a few thousand in the float64 direct-form kernels (LDS window); a
// longer b is cut into segments of `seg` taps,  y[m] = sum_s (b_s * x)[m - s seg]:  segment s is an ordinary filter
// launch over the input shortened by its delay (with as much of the caller's history as it can still see), and its
// result is added onto y from output s seg on.  For .dn the segment length is a multiple of M, so every partial
// result keeps decimation phase 0.
// A handle over taps [t0, t0 + cnt) of `h` on slot `slot`: the ONE place that copies a FIR handle's fields (tap segments, the heads of short calls,
// the per-slot clones of the multi-GPU host path), so that a field added to FirHandle cannot be forgotten in one of them.
static FirHandle *fir_derive(const FirHandle *h, int t0, int cnt, int slot)
{
    const int comp = h->taps_complex ? 2 : 1;
    FirHandle *d = new FirHandle();
    d->kind = H_FIR; d->dtype = h->dtype; d->slot = slot; d->taps_complex = h->taps_complex; d->algo = h->algo; d->wide_out = h->wide_out;
    d->ntaps = cnt;
    d->taps_host.assign(h->taps_host.begin() + (size_t)t0 * comp, h->taps_host.begin() + (size_t)(t0 + cnt) * comp);
    return d;
}

static int fir_part_len(const FirHandle *h)
{
    return dtype_double(h->dtype) ? 2048 : 4096;
}
static bool fir_needs_parts(const FirHandle *h, int L = 1)
{
    // per launch: 4097 taps (float32 overlap-save / direct) or 2049 (float64); an interpolator holds ceil(Ntaps / L) per phase
    const int per_phase = (h->ntaps + L - 1) / L;
    return per_phase > (dtype_double(h->dtype) ? 2049 : 4097);
}
static int fir_dn_any(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int M, void *y_dev, bool scratch_free = true);
static int fir_updn_any(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, int M, void *y_dev, bool scratch_free = true);
static int fir_filter_any(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, void *y_dev);
static int ols_launch_any(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, void *y_dev, int dec = 1);
// y[j] = L sum_t b[(j M mod L) + L t] x[(j M div L) - t] with b cut into segments of `seg` taps, seg a multiple of lcm(L, M):
// segment s delays the up-rate signal by s seg samples = s seg / L input samples = s seg / M outputs, so it is the same
// operation over the input shortened by s seg / L samples, added onto y from output s seg / M on.  With history in front
// of x the segment starts d input samples early (d a multiple of M / gcd(L, M): whole outputs) and lands d L / M outputs earlier.
static int fir_parts_run(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, int M, void *y_dev)
{
    const int g = std::gcd(L, M), lcm = L / g * M, q = M / g;
    const int seg = std::max(fir_part_len(h) * L / lcm, 1) * lcm;   // <= fir_part_len taps per phase
    if (h->part_seg != seg) {
        for (FirHandle *p : h->parts) delete p;
        h->parts.clear();
        for (int t0 = 0; t0 < h->ntaps; t0 += seg) h->parts.push_back(fir_derive(h, t0, std::min(seg, h->ntaps - t0), h->slot));
        h->part_seg = seg;
    }
    for (FirHandle *p : h->parts) p->algo = h->algo;   // (skdsp_fir_set_algo after the parts were made)
    // A segment may start inside the history only by whole output periods (q inputs).  A history that covers a segment's
    // delay is used in full; a shorter one is used up to a multiple of q -- if it is not one itself, the samples
    // x[-n_hist .. -d-1] would be dropped from the outputs just below that segment's first one, so such a call is refused
    // (the host pipeline and the sharded path always hand over a history that is complete or a multiple of q).
    {
        const int64_t last_delay = (int64_t)(h->parts.size() - 1) * seg / L;
        SK_CHECK(q == 1 || n_hist >= last_delay || n_hist % q == 0, SKDSP_ERR_UNSUPPORTED,
                 "fir: %d taps run as %d tap segments; with L/M = %d/%d a partial history (n_hist = %lld < %lld) must be a multiple of %d samples",
                 h->ntaps, (int)h->parts.size(), L, M, (long long)n_hist, (long long)last_delay, q);
    }
    const size_t esz = dtype_size(h->dtype);
    const int scal = dtype_complex(h->dtype) ? 2 : 1;
    hipStream_t s = ctx().stream;
    const int64_t n_out = (n * L) / M;
    void *tmp = nullptr;
    int rc = ws_reserve(2, (size_t)(n_out + 1) * esz + 256, &tmp);
    if (rc) return rc;
    for (size_t si = 0; si < h->parts.size(); ++si) {
        FirHandle *p = h->parts[si];
        const int64_t delay_in = (int64_t)si * seg / L, delay_out = (int64_t)si * seg / M;
        const int64_t d = (std::min(n_hist, delay_in) / q) * q;       // how far this segment starts inside the history
        const int64_t n_s = n - delay_in + d;
        const int64_t off = delay_out - d * L / M;                    // first output this segment contributes to
        const int64_t cnt = std::min((n_s * L) / M, n_out - off);
        if (n_s <= 0 || cnt <= 0) break;
        const char *xs = (const char *)x_dev - (size_t)d * esz;
        void *dst = si == 0 ? y_dev : tmp;
        if (L == 1 && M == 1)
            rc = fir_algo_for(p, n_s) == SKDSP_FIR_OLS ? ols_launch_any(p, xs, n_s, n_hist - d, dst)
                                                        : fir_direct_launch(p, xs, n_s, n_hist - d, 1, 1, n_s, dst, s);
        else if (L == 1)
            rc = fir_dn_any(p, xs, n_s, n_hist - d, M, dst, false);   // (workspace slot 2 is `tmp` -- possibly `dst` -- here)
        else
            rc = fir_updn_any(p, xs, n_s, n_hist - d, L, M, dst, false);   // (workspace slot 2 is `tmp` here: no scratch-using forms; writes (n_s L) / M <= n_out outputs)
        if (rc) return rc;
        if (si > 0 && (rc = accumulate_launch((char *)y_dev + (size_t)off * esz, tmp, cnt * scal, dtype_double(h->dtype), s))) return rc;
    }
    return SKDSP_OK;
}

static int ols_launch_any(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, void *y_dev, int dec)
{
    if (dtype_double(h->dtype)) return fir_ols64_launch(h, x_dev, n, n_hist, y_dev, ctx().stream, dec);
    return fir_ols_launch(h, x_dev, n, n_hist, y_dev, ctx().stream, dec);
}

// scratch_free: workspace slot 2 may hold the full-rate result of the last-resort path (false inside fir_parts_run, which holds it: slot 3 then --
// the planes of a complex IIR call, never alive during a FIR call)
static int fir_dn_any(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int M, void *y_dev, bool scratch_free)
{
    const int full_slot = scratch_free ? 2 : 3;
    if (fir_needs_parts(h)) return fir_parts_run(h, x_dev, (n / M) * M, n_hist, 1, M, y_dev);
    if (dtype_double(h->dtype)) {  // float64: the decimating overlap-save store beats Ntaps / M direct FP64 taps per kept sample early
        if (M > 1 && pick_fir_algo(h, n) == SKDSP_FIR_OLS && !opt().dn_no_ols && h->ntaps / M >= 24)
            return fir_ols64_launch(h, x_dev, n, n_hist, y_dev, ctx().stream, M);
        int rc = fir_direct_launch(h, x_dev, n, n_hist, 1, M, n / M, y_dev, ctx().stream);
        if (rc == SKDSP_ERR_UNSUPPORTED && M > 1) {   // (a stride the polyphase kernels' LDS window does not hold: see below)
            if (fir_ols64_supported(h) && !opt().dn_no_ols) return fir_ols64_launch(h, x_dev, n, n_hist, y_dev, ctx().stream, M);
            void *full = nullptr;
            const int64_t nk = (n / M) * M;
            if ((rc = ws_reserve(full_slot, (size_t)nk * dtype_size(h->dtype) + 256, &full))) return rc;
            if ((rc = fir_filter_any(h, x_dev, nk, n_hist, full))) return rc;
            return downsample_launch(full, nk, M, 0, h->dtype, y_dev, ctx().stream);
        }
        return rc;
    }
    bool ols = M > 1 && fir_ols_supported(h) && pick_fir_algo(h, n) == SKDSP_FIR_OLS && !opt().dn_no_ols;
    const bool fold = M % 2 == 0 && opt().fir_dn_fold;   // even M: the overlap-save tile transforms only the kept outputs back (ols_fold_kernel)
    if (ols) {
        // Which engine (profiles/r05/fir_dn.txt, 2^26 inputs).  The matrix-pipe kernel computes kept outputs only and costs with the taps per kept
        // output u = Ntaps / M; the overlap-save tile costs the same whatever the filter: with the folded inverse transform 0.155 - 0.19 ms
        // (complex64; float32 0.085 - 0.105), with the decimating store (odd M) the plain filter's 0.21 - 0.23.  Measured crossovers: complex64
        // M = 4 from the shortest filter overlap-save takes, M = 2 from u = 96, M = 8, 12, 16 from u = 64, M = 6, 10 from u = 128; float32 from
        // u = 128 (M = 2: 192).  Where the matrix-pipe kernel does not cover the shape (complex taps, lag ranges beyond its 48 blocks) the
        // register sliding-window kernel is the alternative, and cheaper below a few dozen taps per kept output.
        const int kb = h->algo == SKDSP_FIR_OLS ? -1 : fir_bx_blocks(h, 1, M);
        const int u = h->ntaps / M;
        const bool f32 = h->dtype == SKDSP_F32;
        if (kb < 0) ols = true;                                                  // (forced by the caller)
        else if (kb == 0) ols = u >= (f32 ? 64 : 24);
        // (end of round 6, with the matrix-pipe kernel's paired column tiles: complex64 M = 16, u = 64 0.146 against 0.162 ms; float32 M = 8, u = 128 0.093 / 0.098)
        else if (fold) ols = u >= (f32 ? (M == 2 ? 192 : (M == 4 ? 128 : 160)) : (M == 4 ? 0 : (M == 2 ? 96 : (M % 16 == 0 ? 96 : (M % 4 == 0 ? 64 : 128)))));
        else ols = M <= 4 && kb > 12;                                            // (M = 3: complex64 512 taps 0.256 ms against 0.215, float32 0.132 / 0.100)
    }
    // M = 3: the frequency-domain decimator (fir_dn4k.hip: M forward transforms accumulated, ONE inverse per tile of kept outputs) wherever the
    // decimating store would run; even M: the folded inverse is ahead of it everywhere (M = 2, 1024 taps: 0.189 against 0.219 ms; M = 4: 0.174 /
    // 0.237; float32 0.097 / 0.116).  Option fir_dn4k = 2: wherever it applies (A/B timing, tests)
    if (M > 1 && opt().fir_dn4k && fir_dn4k_supported(h, M) && n / M >= 2048 &&
        (opt().fir_dn4k >= 2 || (ols && !fold && (h->dtype == SKDSP_F32 || h->ntaps > 1536))))
        return fir_dn4k_launch(h, x_dev, n, n_hist, M, y_dev, ctx().stream);
    if (ols) return fir_ols_launch(h, x_dev, n, n_hist, y_dev, ctx().stream, M);
    int rc = fir_direct_launch(h, x_dev, n, n_hist, 1, M, n / M, y_dev, ctx().stream);
    if (rc == SKDSP_ERR_UNSUPPORTED && M > 1) {
        // a stride the polyphase kernels' LDS window does not hold (a few hundred taps and M in the thousands): the decimating
        // overlap-save store takes any M; without that engine, the full-rate filter and a strided copy
        // (that store's index arithmetic is exact up to M = 32768 -- fir_ols_launch checks it --: beyond, the full-rate filter and the strided copy)
        if (fir_ols_supported(h) && !opt().dn_no_ols && M <= 32768) return fir_ols_launch(h, x_dev, n, n_hist, y_dev, ctx().stream, M);
        void *full = nullptr;
        const int64_t nk = (n / M) * M;
        if ((rc = ws_reserve(full_slot, (size_t)nk * dtype_size(h->dtype) + 256, &full))) return rc;
        if ((rc = fir_filter_any(h, x_dev, nk, n_hist, full))) return rc;
        return downsample_launch(full, nk, M, 0, h->dtype, y_dev, ctx().stream);
    }
    return rc;
}

// .up / fused L over M: one polyphase launch, or tap segments when a phase holds more taps than a launch takes
// multirate_FIR.up: polyphase kernels or the overlap-save walk over (tile, phase) pairs (fir_ols.hip)?  Both are timed models of this
// board at 2^26 outputs (tools/time_fir_up.py; ms), scaled to the call: the polyphase kernels cost per tap of a phase -- little where the
// matrix-pipe kernel covers the shape, 3-4x that where it does not -- the walk costs per (tile, phase) pair whatever the phase length,
// plus what its stride-L stores cost, and runs in rounds of one pair per resident workgroup.
// multirate_FIR.up through the overlap-save walk: from which L on the phases leave as rows of scratch and a second kernel weaves them
// (measured crossovers of profiles/r03/fir_up.txt -- the walk serves float64 and > 1025 taps per phase today; 16-byte samples never: their strided stores are full-width requests already)
static bool fir_up_rows(const FirHandle *h, int L, bool paired = false)
{
    const int o = opt().fir_up_rows_min;
    if (o == 0) return false;
    if (o > 0) return L >= o;
    if (paired) return !dtype_double(h->dtype) && L % 2 == 0 && L / 2 >= 7;   // (8-byte pairs: the complex64 crossover, in phases; 16-byte pairs never;
                                                                              //  odd L in pairs: the strided form only)
    switch (h->dtype) {
    case SKDSP_F32: return L >= 9;
    case SKDSP_C64: return L >= 7;
    case SKDSP_F64: return L >= 6;
    default: return false;
    }
}

// The one-workgroup-per-input-tile interpolators (fir_up4k.hip: up to four passes per thread; fir_up2k.hip: all passes of a row per
// thread): which one a call takes (0: none applies), and what it costs in ms per 2^26 outputs on this board (round-4 timings,
// tools/time_up4k.py; the measured shapes had 3 - 6 % of their tile in the overlap, so the figure is scaled to the call's overlap).
static int fir_up_tile_kind(const FirHandle *h, int L)
{
    if (dtype_double(h->dtype) || !opt().fir_up4k) return 0;
    const int passes = h->dtype == SKDSP_F32 ? (L + 1) / 2 : L;   // (float32: two phases per complex pass)
    // float32, L = 2: one pass -- the walk's pair form IS the plain filter's 8192-point tile with an 8-byte store, and stays ahead of the
    // 4096-point tile (512 / 1024 taps per phase: 0.117 / 0.125 ms against 0.126 / 0.140)
    if (passes == 1 && opt().fir_up4k < 2) return 0;
    if (opt().fir_up2k && fir_up2k_supported(h, L) && (opt().fir_up2k >= 2 || passes > 4)) return 2;
    return fir_up4k_supported(h, L) ? 4 : 0;
}
static double fir_up_tile_ms(const FirHandle *h, int L, int kind, int *V_out)
{
    const int T = (h->ntaps + L - 1) / L;
    const int passes = h->dtype == SKDSP_F32 ? (L + 1) / 2 : L;
    const bool cplx = h->dtype == SKDSP_C64;
    double ms;
    int N, ov;
    if (kind == 4) {   // 4096-point tile, groups of four passes: one group is one burst per row, more are pieces written far apart
        N = 4096; ov = std::max(256, (T - 1 + 255) / 256 * 256);
        // (one group: the forward transform is shared by `passes` inverse ones -- 0.2025 / 0.199 / 0.187 ms at 2 / 3 / 4 complex64 passes,
        // 0.1225 / 0.0946 / 0.105 / 0.096 at 1 .. 4 float32 passes, profiles/r04/fir_up.txt)
        static const double c4[5] = {0.0, 0.26, 0.2025, 0.199, 0.187}, f4[5] = {0.0, 0.1225, 0.0946, 0.105, 0.096};
        ms = cplx ? (passes <= 4 ? c4[passes] : 0.30 + 0.012 * std::min(passes, 12)) : (passes <= 4 ? f4[passes] : 0.10 + 0.005 * std::min(passes, 12));
        ms *= (4096.0 - 256.0) / 4096.0;
    } else {           // 2048-point tile, up to twelve passes per thread
        N = 2048; ov = std::max(64, (T - 1 + 63) / 64 * 64);
        if (passes <= 12) ms = cplx ? 0.19 + 0.0025 * passes : 0.085 + 0.0035 * passes;
        else ms = cplx ? 0.36 : 0.16;
        if (passes % 2) ms *= 1.07;   // (an odd row: every lane stores its own pieces)
        ms *= (2048.0 - 64.0) / 2048.0;
    }
    *V_out = N - ov;
    return ms * (double)N / (double)(N - ov);
}

// multirate_FIR.up, even L, on tiles of the OUTPUT (fir_ols.hip: ols_rep_kernel): ms per 2^26 outputs (round-5 timings, profiles/r05/fir_up.txt: the plain
// filter's tile with a quarter of its forward transform and 1 / L of its loads; the overlap is that of the WHOLE filter at the high rate)
static double fir_up_rep_ms(const FirHandle *h, int L)
{
    const int ov = std::max(512, (h->ntaps - 1 + 511) / 512 * 512);
    const bool cplx = h->dtype == SKDSP_C64;
    const bool pow2 = (L & (L - 1)) == 0 && L <= 16;   // (else the decimated grid is itself zero-stuffed: the guarded loader, 4-byte samples feel it)
    const double base = cplx ? (L == 2 ? 0.161 : 0.152) : (L == 2 ? 0.087 : (L == 4 ? 0.083 : 0.0885)) * (pow2 ? 1.0 : 1.18);
    return base * 8192.0 / (8192.0 - ov);
}

// best: which frequency-domain engine the model found cheapest (1 the walk over (tile, phase) pairs, 2 an input-tile interpolator, 3 the output-tile one)
static bool fir_up_prefers_ols(const FirHandle *h, int L, int64_t n, int M = 1, int *best = nullptr)
{
    if (best) *best = 1;
    const int T = (h->ntaps + L - 1) / L;
    const int floor_t = opt().fir_up_ols_min;   // < 0: wherever supported from -floor_t taps per phase on, no cost model (tests, A/B timing)
    const bool dbl = dtype_double(h->dtype);
    int floor_eff = std::abs(floor_t);   // (many phases: the polyphase kernels lose their reuse early -- let the cost model see shorter phases too)
    if (floor_t > 0 && L > 64) floor_eff = std::max(8, floor_t / 8);
    else if (floor_t > 0 && L > 16) floor_eff = std::max(8, floor_t / 4);
    if (M == 1 && floor_t > 0 && fir_up_tile_kind(h, L)) floor_eff = std::min(floor_eff, 24);   // (the tile interpolators cross over with the polyphase kernels at short phases already)
    if (floor_t == 0 || T < floor_eff || n < 8192 || !(dbl ? fir_ols64_up_supported(h, L) : fir_ols_up_supported(h, L))) return false;
    if (M > 1 && L > 64) return false;   // (the every-M-th store's exact-division range; the scratch + copy form is not worth it there)
    if (floor_t < 0) return true;
    if ((opt().fir_algo != SKDSP_FIR_AUTO ? opt().fir_algo : h->algo) == SKDSP_FIR_DIRECT) return false;
    if (fir_needs_parts(h, L)) return true;   // (longer than one polyphase launch takes)
    // all figures: ms per 2^26 up-rate samples on this board (the walk and the float64 kernels: profiles/r03/fir_up.txt, fir_updn.txt; the matrix-pipe and tile kernels: profiles/r04)
    const double Lf = (double)L;
    const bool cplx = dtype_complex(h->dtype);
    double ols, base, poly, copy;   // base: the walk without what its stride-L stores cost
    int V;
    if (dbl) {   // FP64 direct taps against the float64 walk (4096-point tiles)
        base = cplx ? 0.42 : 0.26;
        ols = cplx ? 0.60 + 0.008 * std::min(Lf, 24.0) : 0.29 + 0.02 * std::min(Lf, 12.0);
        poly = cplx ? 0.5 + 0.0055 * T : (T <= 128 ? 0.17 + 0.0018 * T : 0.1 + 0.0028 * T);
        if (cplx && L > 16) poly = std::max(poly, 1.0);   // (measured 1.02 ... 1.12 from L = 24 on, whatever the phase length)
        if (T > 128) poly *= std::max(1.0, Lf / 4.0);   // (many long phases: the tap tables fall out of the cache)
        else if (L > 16 && !cplx) poly *= 1.0 + Lf / 12.0;
        copy = cplx ? 0.20 : 0.10;
        V = 4096 - ((T - 1 + 255) / 256) * 256;
    } else {
        int bx_rt = 0;
        const int bx_kb = fir_bx_blocks(h, L, M, &bx_rt);   // (the matrix-pipe polyphase kernel covers the shape: its time goes with its 32-lag blocks)
        const bool bx = bx_kb > 0;
        base = cplx ? 0.23 : 0.125;
        ols = cplx ? 0.27 + 0.022 * std::min(Lf, 20.0) : 0.13 + 0.018 * std::min(Lf, 28.0);
        // profiles/r04/fir_up.txt (fp16 pieces): complex64 0.106 - 0.122 up to 3 blocks, then + 0.0145 per block (5: 0.13, 7: 0.165; one row tile, L = 2:
        // 0.122 / 0.127 / 0.143 / 0.159 / 0.194 / 0.223 for 2 / 3 / 4 / 5 / 7 / 9); float32 0.080 - 0.096 up to 5 blocks, 0.099 at 7 (L = 2: 0.075 ... 0.133)
        if (bx && cplx) poly = bx_rt == 1 ? 0.093 + 0.0145 * bx_kb : std::max(L >= 8 ? 0.118 : 0.106, 0.062 + 0.0145 * bx_kb);
        else if (bx) poly = bx_rt == 1 ? 0.058 + 0.0084 * bx_kb : std::max(0.081 * (L > 8 ? 1.15 : (L == 8 ? 1.06 : 1.0)), 0.04 + 0.0084 * bx_kb);
        else poly = cplx ? 0.02 + 0.0037 * T : 0.03 + 0.0018 * T;
        if (!bx && L > 8 && L <= 16) poly *= 1.0 + 0.05 * (Lf - 8.0);   // (48 taps per phase: 0.116 modelled, 0.1395 measured at L = 12)
        if (!bx && T > 256) poly *= std::max(1.0, Lf / 4.0);
        else if (L > 16 && !bx) poly *= 1.0 + Lf / 12.0;   // (one tap table per phase: the polyphase kernels lose their reuse)
        copy = cplx ? 0.10 : 0.06;
        V = 8192 - ((T - 1 + 511) / 512) * 512;
    }
    if (M == 1 && fir_up_rows(h, L)) ols = std::min(ols, dbl ? 0.45 : (cplx ? 0.45 : 0.245));   // (rows + weave: whatever L is)
    if (M == 1 && !dbl && fir_ols_up_pairs(h, L, 1, nullptr))   // float32, even L: L / 2 complex passes per tile of real input, 8-byte outputs
        ols = std::min(0.11 + 0.007 * Lf, 0.235);
    if (M == 1 && dbl && fir_ols64_up_pairs(h, L, 1, nullptr))   // float64 likewise, 16-byte outputs
        ols = 0.25 + 0.005 * std::min(Lf, 16.0);
    if (M > 1) {   // L / M: the polyphase kernels compute the kept outputs only; the walk computes all and stores (or copies) every M-th
        poly /= (double)M;
        if (M <= 4096 && opt().fir_updn_fused) ols = base + (ols - base) / (double)M;
        else ols += copy;
    }
    // the walk runs in rounds of one (tile, phase) pair per resident workgroup; the polyphase kernels scale with the length
    const double slots = 2.0 * ctx().num_cus;
    const double pairs = (double)((n + V - 1) / V) * (cplx ? 1.0 : 0.5) * Lf;
    ols *= std::ceil(pairs / slots) * slots * (double)V * (cplx ? 1.0 : 2.0) / 67108864.0;
    poly *= (double)n * Lf / 67108864.0;
    if (M == 1) {   // the tile interpolators replace the walk wherever they apply: rounds of one INPUT tile (all phases) per resident workgroup
        const int kind = fir_up_tile_kind(h, L);
        if (kind) {
            int Vt = 0;
            const double ms = fir_up_tile_ms(h, L, kind, &Vt);
            const double tiles = (double)((n + Vt - 1) / Vt);
            const double tms = ms * std::ceil(tiles / slots) * slots * (double)Vt * Lf / 67108864.0;
            if (tms < ols) { ols = tms; if (best) *best = 2; }
        }
        if (opt().fir_up_rep && fir_ols_rep_supported(h, L)) {
            const double rms = fir_up_rep_ms(h, L) * (double)n * Lf / 67108864.0;
            if (rms < ols) { ols = rms; if (best) *best = 3; }
        }
    }
    return ols < poly;
}

// scratch_free: workspace slot 2 may be used (rows of the .up walk, the unfused L / M copy); false inside fir_parts_run, which holds it
static int fir_updn_any(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, int M, void *y_dev, bool scratch_free)
{
    if (L == 1) return fir_dn_any(h, x_dev, n, n_hist, M, y_dev);
    auto walk = [&](void *out, int dec) {
        return dtype_double(h->dtype) ? fir_ols64_up_launch(h, x_dev, n, n_hist, L, out, ctx().stream, dec)
                                      : fir_ols_up_launch(h, x_dev, n, n_hist, L, out, ctx().stream, dec);
    };
    // even L: tiles of the OUTPUT, the zero-stuffed tile's spectrum from its non-zero columns (ols_rep_kernel); option fir_up_rep = 2: wherever it applies
    if (M == 1 && n * L >= 8192 && fir_ols_rep_supported(h, L)) {
        int best = 0;
        if (opt().fir_up_rep >= 2 || (opt().fir_up4k < 2 && opt().fir_up_ols_min > 0 && fir_up_prefers_ols(h, L, n, 1, &best) && best == 3))   // (an engine forced by option stays forced)
            return fir_ols_rep_launch(h, x_dev, n, n_hist, L, y_dev, ctx().stream);
    }
    // one workgroup per input tile, all L phases from ONE forward transform (fir_up4k.hip / fir_up2k.hip); option fir_up4k: 0 never, 2
    // wherever one applies (tests, A/B timing), 1 where the cost model above prefers the frequency domain
    if (M == 1 && n >= 2048) {
        const int kind = fir_up_tile_kind(h, L);
        if (kind && (opt().fir_up4k >= 2 || fir_up_prefers_ols(h, L, n)))
            return kind == 2 ? fir_up2k_launch(h, x_dev, n, n_hist, L, y_dev, ctx().stream) : fir_up4k_launch(h, x_dev, n, n_hist, L, y_dev, ctx().stream);
    }
    if (M == 1 && fir_up_prefers_ols(h, L, n)) {
        const bool dbl = dtype_double(h->dtype);
        bool paired = dbl ? fir_ols64_up_pairs(h, L, 1, y_dev) : fir_ols_up_pairs(h, L, 1, y_dev);
        bool rows = scratch_free && fir_up_rows(h, L, paired) && !(paired && L == 2);   // (one pair is one row: nothing to weave)
        if (rows && paired && L % 2) {   // an odd L in pairs has the strided form only: rows asked for by option win, else the pairs
            if (opt().fir_up_rows_min > 0) paired = false; else rows = false;
        }
        if (rows) {
            // many phases: an output stored between outputs of other phases is a write request of its own, so the phases leave as rows
            // with the plain filter's stores and interleave_launch weaves them (one more pass over the output, still cheaper from L = 6 ... 9 on)
            const int rows_n = paired ? L / 2 : L;
            const int row_dtype = paired ? (dbl ? SKDSP_C128 : SKDSP_C64) : h->dtype;
            const int64_t pitch = (int64_t)round_up((size_t)n, 64);
            void *rows = nullptr;
            int rc = ws_reserve(2, (size_t)pitch * rows_n * dtype_size(row_dtype) + 256, &rows);
            if (rc) return rc;
            rc = dbl ? fir_ols64_up_launch(h, x_dev, n, n_hist, L, rows, ctx().stream, 1, pitch, paired)
                     : fir_ols_up_launch(h, x_dev, n, n_hist, L, rows, ctx().stream, 1, pitch, paired);
            if (rc) return rc;
            return interleave_launch(rows, n, rows_n, pitch, row_dtype, y_dev, ctx().stream);
        }
        if (paired) return dbl ? fir_ols64_up_launch(h, x_dev, n, n_hist, L, y_dev, ctx().stream, 1, 0, 1) : fir_ols_up_launch(h, x_dev, n, n_hist, L, y_dev, ctx().stream, 1, 0, 1);
        return walk(y_dev, 1);
    }
    if (M > 1 && (scratch_free || (M <= 4096 && opt().fir_updn_fused)) && fir_up_prefers_ols(h, L, n, M)) {   // long phases: all n L outputs by the walk, every M-th of them kept
        if (M <= 4096 && opt().fir_updn_fused) return walk(y_dev, M);   // ... by its store
        void *full = nullptr;                                           // ... or out of scratch
        int rc = ws_reserve(2, (size_t)n * L * dtype_size(h->dtype) + 256, &full);
        if (rc) return rc;
        if ((rc = walk(full, 1))) return rc;
        return downsample_launch(full, n * L, M, 0, h->dtype, y_dev, ctx().stream);
    }
    if (fir_needs_parts(h, L)) return fir_parts_run(h, x_dev, n, n_hist, L, M, y_dev);
    int rc = fir_direct_launch(h, x_dev, n, n_hist, L, M, (n * L) / M, y_dev, ctx().stream);
    if (rc == SKDSP_ERR_UNSUPPORTED && M > 1 && M <= 4096 && L <= 64 && opt().fir_up_ols_min != 0 &&
        (dtype_double(h->dtype) ? fir_ols64_up_supported(h, L) : fir_ols_up_supported(h, L)))
        return walk(y_dev, M);   // (a stride the polyphase kernels' LDS window does not hold)
    return rc;
}

// A call from rest over n < Ntaps samples computes y[m] = sum_(k <= m) b[k] x[m - k], m < n: taps b[n ...] are never reached.  It runs on a
// copy of the filter cut to the next power of two >= n (at most log2(Ntaps) copies per handle) -- the same outputs exactly, less work, and
// the rounding of a float32 engine (a few 1e-8 of sum |b| max |x|: the FFT engines round against the WHOLE filter) shrinks with the taps
// that matter.  (Found by the differential test at north_star's bound without its former factor 2: 17 rows of 100 samples through a
// 1024-tap low-pass, whose first 100 taps are its tail -- 1.7e-6 of the tiny start-up transient before, far inside 1e-6 after.)
static FirHandle *fir_head(FirHandle *h, int64_t n)
{
    if (n >= h->ntaps || n < 1) return h;
    int keep = 1;
    while (keep < n) keep <<= 1;
    if (keep >= h->ntaps) return h;
    for (FirHandle *t : h->heads)
        if (t->ntaps == keep) { t->algo = h->algo; return t; }
    // (at most log2(Ntaps) <= 13 heads per handle -- one per power of two below the tap count -- each with the tables of the engines it has run on: they
    // live as long as the handle)
    FirHandle *t = fir_derive(h, 0, keep, h->slot);
    h->heads.push_back(t);
    return t;
}

static int fir_filter_any(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, void *y_dev)
{
    if (n_hist == 0 && n < h->ntaps) h = fir_head(h, n);
    if (fir_needs_parts(h)) return fir_parts_run(h, x_dev, n, n_hist, 1, 1, y_dev);
    if (pick_fir_algo(h, n) == SKDSP_FIR_OLS) return ols_launch_any(h, x_dev, n, n_hist, y_dev);
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
    return sync_checked();
}


// ---------------------------------------------------------------------------------------------------------------
// Host-pointer entry points on LONG vectors: chunk pipeline.
//
// The reference call hands over a NumPy array and expects one back (multirate_helper.py:104-127, 169-192), so the
// drop-in path crosses PCIe twice: 2 x 2.4 ms per 128 MiB against 0.06 ms of kernel.  Staging the whole vector,
// filtering it and copying it back one after the other leaves each PCIe direction idle half of the time.  Here the
// vector is cut into chunks of 2^host_chunk_log2 samples that are exact continuations of each other (FIR: the chunk's
// copy starts Ntaps-1 samples early and the kernel gets them as n_hist; IIR: zi / zf), and three things run at once:
//   the caller's thread   H2D of chunk k+1 (pageable source: the runtime's own staging runs at the link rate) and the
//                         launches of chunk k (compute stream waits for the copy's event)
//   a helper thread       D2H of chunk k-1 into the caller's result array (the other direction of the link)
// with two device buffers per direction.  With several slots bound (skdsp_init_devices: one per GPU) the chunks of a FIR
// are dealt to all of them -- each slot runs this pipeline over a contiguous range of chunks from its own worker thread
// and over its own PCIe link; the history of a range's first chunk comes from the host vector like any other chunk's,
// so the GPUs exchange nothing.
struct HostPipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t in_ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    void *din[2] = {nullptr, nullptr}, *dout[2] = {nullptr, nullptr};
    size_t cap_in = 0, cap_out = 0;
};

static void pipe_free(Context &c)
{
    HostPipe *p = c.pipe;
    if (!p) return;
    for (int i = 0; i < 2; ++i) {
        if (p->din[i]) (void)hipFree(p->din[i]);
        if (p->dout[i]) (void)hipFree(p->dout[i]);
        if (p->in_ready[i]) (void)hipEventDestroy(p->in_ready[i]);
        if (p->done[i]) (void)hipEventDestroy(p->done[i]);
    }
    if (p->s_in) (void)hipStreamDestroy(p->s_in);
    if (p->s_out) (void)hipStreamDestroy(p->s_out);
    delete p;
    c.pipe = nullptr;
}

static int pipe_ensure(Context &c, size_t in_bytes, size_t out_bytes)
{
    if (!c.pipe) {
        HostPipe *p = new HostPipe();
        c.pipe = p;
        SK_HIP(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
        SK_HIP(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            SK_HIP(hipEventCreateWithFlags(&p->in_ready[i], hipEventDisableTiming));
            SK_HIP(hipEventCreateWithFlags(&p->done[i], hipEventDisableTiming));
        }
    }
    HostPipe *p = c.pipe;
    if (in_bytes > p->cap_in) {
        SK_HIP(hipStreamSynchronize(c.stream));
        for (int i = 0; i < 2; ++i) {
            if (p->din[i]) SK_HIP(hipFree(p->din[i]));
            p->din[i] = nullptr;
        }
        p->cap_in = 0;
        for (int i = 0; i < 2; ++i) SK_HIP(hipMalloc(&p->din[i], in_bytes));
        p->cap_in = in_bytes;
    }
    if (out_bytes > p->cap_out) {
        SK_HIP(hipStreamSynchronize(c.stream));
        for (int i = 0; i < 2; ++i) {
            if (p->dout[i]) SK_HIP(hipFree(p->dout[i]));
            p->dout[i] = nullptr;
        }
        p->cap_out = 0;
        for (int i = 0; i < 2; ++i) SK_HIP(hipMalloc(&p->dout[i], out_bytes));
        p->cap_out = out_bytes;
    }
    return SKDSP_OK;
}

// how a long vector is cut: chunk k covers inputs [k C, min((k+1) C, n)) and outputs [(k C L) / M, (end L) / M)
struct ChunkPlan {
    int64_t n = 0, C = 0, nchunks = 0, hist = 0;
    int L = 1, M = 1;
    size_t esz = 0;       // bytes per input / output sample on the device
    bool wide = false;    // results leave as float64 / complex128 (twice esz on the host side)
    int64_t in_begin(int64_t k) const { return k * C; }
    int64_t in_end(int64_t k) const { return std::min<int64_t>((k + 1) * C, n); }
    int64_t hist_of(int64_t k) const { return std::min<int64_t>(hist, k * C); }
    int64_t out_begin(int64_t k) const { return (in_begin(k) * L) / M; }
    int64_t out_end(int64_t k) const { return k + 1 == nchunks ? (n * L) / M : (in_end(k) * L) / M; }
};

// the planner (also exported for the CPU tests: skdsp_host_chunk_plan)
static ChunkPlan plan_chunks(int64_t n, int L, int M, int64_t hist, size_t esz, bool wide, int chunk_log2)
{
    ChunkPlan p;
    p.n = n; p.L = L; p.M = M; p.hist = hist; p.esz = esz; p.wide = wide;
    int64_t C = (int64_t)1 << std::max(10, std::min(chunk_log2, 30));
    C = std::max<int64_t>(C / M, 1) * M;       // chunk starts stay multiples of M: output phase 0 stays aligned
    if (C < hist) C = ((hist + M - 1) / M) * M;  // (keeps the staging buffers within twice a chunk)
    p.C = C;
    p.nchunks = std::max<int64_t>((n + C - 1) / C, 1);
    return p;
}

typedef int (*chunk_kernel_fn)(void *self, const void *x_dev, int64_t n_k, int64_t n_hist, void *y_dev, int64_t k);

// chunks [k0, k1) of the plan on the CURRENT slot; x / y: the caller's whole host vectors
static int run_pipeline(const ChunkPlan &p, int64_t k0, int64_t k1, const char *x, char *y, chunk_kernel_fn kern, void *self)
{
    Context &c = ctx();
    if (k1 <= k0) return SKDSP_OK;
    const size_t esz = p.esz, esz_out = p.wide ? 2 * esz : esz;
    const size_t in_cap = (size_t)(p.C + p.hist) * esz + kHeadroomBytes + 512;
    const size_t out_cap = (size_t)((p.C * p.L) / p.M + 2) * esz_out + 512;
    int rc = pipe_ensure(c, in_cap, out_cap);
    if (rc) return rc;
    HostPipe *hp = c.pipe;
    void *narrow = nullptr;
    if (p.wide && (rc = ws_reserve(1, (size_t)((p.C * p.L) / p.M + 2) * esz + 256, &narrow))) return rc;

    std::mutex mu;
    std::condition_variable cv;
    int64_t posted = k0, drained = k0;   // chunks handed to / finished by the copy-back thread
    bool abort_flag = false;
    int helper_rc = SKDSP_OK;
    char helper_err[256] = "";
    const int device = c.device;
    std::thread helper([&]() {
        if (hipSetDevice(device) != hipSuccess) {
            std::lock_guard<std::mutex> lk(mu);
            helper_rc = SKDSP_ERR_HIP;
            snprintf(helper_err, sizeof(helper_err), "host pipeline: hipSetDevice(%d) failed in the copy-back thread", device);
            drained = k1;
            cv.notify_all();
            return;
        }
        for (int64_t k = k0; k < k1; ++k) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return posted > k || abort_flag; });
                if (abort_flag && posted <= k) break;
            }
            const int b = (int)((k - k0) & 1);
            const size_t bytes = (size_t)(p.out_end(k) - p.out_begin(k)) * esz_out;
            hipError_t e = hipEventSynchronize(hp->done[b]);
            if (e == hipSuccess && bytes)
                e = hipMemcpyAsync(y + (size_t)p.out_begin(k) * esz_out, hp->dout[b], bytes, hipMemcpyDeviceToHost, hp->s_out);
            if (e == hipSuccess) e = hipStreamSynchronize(hp->s_out);
            std::lock_guard<std::mutex> lk(mu);
            if (e != hipSuccess && helper_rc == SKDSP_OK) {
                helper_rc = SKDSP_ERR_HIP;
                snprintf(helper_err, sizeof(helper_err), "host pipeline: copy back of chunk %lld failed: %s", (long long)k, hipGetErrorString(e));
            }
            drained = k + 1;
            cv.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        drained = k1;
        cv.notify_all();
    });

    auto body = [&]() -> int {
        for (int64_t k = k0; k < k1; ++k) {
            const int b = (int)((k - k0) & 1);
            const int64_t hk = p.hist_of(k), ib = p.in_begin(k), nk = p.in_end(k) - ib;
            // din[b] was last read by the kernels of chunk k-2; dout[b] was last read by the copy back of chunk k-2
            if (k - k0 >= 2) {
                SK_HIP(hipEventSynchronize(hp->done[b]));
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return drained >= k - 1 || helper_rc != SKDSP_OK; });
                if (helper_rc != SKDSP_OK) return helper_rc;
            }
            char *xd = (char *)hp->din[b] + kHeadroomBytes + (size_t)p.hist * esz;
            xd = (char *)(((uintptr_t)xd + 255) & ~(uintptr_t)255);   // x[0] of the chunk 256-byte aligned, history in front of it
            SK_HIP(hipMemcpyAsync(xd - (size_t)hk * esz, x + (size_t)(ib - hk) * esz, (size_t)(nk + hk) * esz, hipMemcpyHostToDevice, hp->s_in));
            SK_HIP(hipEventRecord(hp->in_ready[b], hp->s_in));
            SK_HIP(hipStreamWaitEvent(c.stream, hp->in_ready[b], 0));
            const int64_t n_out = p.out_end(k) - p.out_begin(k);
            void *yd = p.wide ? narrow : hp->dout[b];
            int r = kern(self, xd, nk, hk, yd, k);
            if (r) return r;
            if (p.wide && n_out > 0 && (r = widen_launch(narrow, (int64_t)((size_t)n_out * esz / 4), hp->dout[b], c.stream))) return r;
            SK_HIP(hipEventRecord(hp->done[b], c.stream));
            {
                std::lock_guard<std::mutex> lk(mu);
                posted = k + 1;
            }
            cv.notify_all();
        }
        return SKDSP_OK;
    };
    rc = body();
    {
        std::lock_guard<std::mutex> lk(mu);
        if (rc) abort_flag = true;
    }
    cv.notify_all();
    helper.join();
    (void)hipStreamSynchronize(c.stream);
    if (rc) return rc;
    if (helper_rc) {
        set_error("%s", helper_err);
        return helper_rc;
    }
    return async_err_check(c);
}

// Deal the chunks of a plan to every bound slot (contiguous ranges); make_self(slot) gives the per-slot kernel argument
// (the handle's clone on that slot).  One worker thread per extra slot; the caller's thread serves its own slot.
static thread_local int tl_slot_limit = 0;   // > 0: this call uses at most that many slots (skdsp_fir_filter_sharded)

static int run_on_slots(const ChunkPlan &p, const char *x, char *y, chunk_kernel_fn kern, void *(*make_self)(void *, int), void *base_self,
                        bool allow_multi)
{
    const int home = ctx().slot;
    int nslots = allow_multi && opt().host_multi_slot ? slot_count() : 1;
    if (tl_slot_limit > 0 && nslots > tl_slot_limit) nslots = tl_slot_limit;
    if (nslots > p.nchunks) nslots = (int)p.nchunks;
    if (nslots <= 1) return run_pipeline(p, 0, p.nchunks, x, y, kern, make_self(base_self, home));
    std::vector<void *> selfs((size_t)nslots, nullptr);
    std::vector<int> slots;
    slots.push_back(home);
    for (int s = 0; s < slot_count() && (int)slots.size() < nslots; ++s)
        if (s != home && ctx_of(s).ready) slots.push_back(s);
    nslots = (int)slots.size();
    for (int i = 0; i < nslots; ++i) {
        selfs[i] = make_self(base_self, slots[i]);
        if (!selfs[i]) return SKDSP_ERR_NOMEM;
    }
    std::vector<int> rcs((size_t)nslots, SKDSP_OK);
    std::vector<std::string> errs((size_t)nslots), paths((size_t)nslots);   // (paths: the engines each worker thread launched -- the record is thread-local)
    std::vector<std::thread> workers;
    auto range = [&](int i, int64_t &a, int64_t &b) {
        a = p.nchunks * i / nslots;
        b = p.nchunks * (i + 1) / nslots;
    };
    for (int i = 1; i < nslots; ++i) {
        workers.emplace_back([&, i]() {
            int r = select_slot(slots[i]);
            if (!r) {
                std::lock_guard<std::mutex> lk(ctx().mu);   // the slot's own lock: other callers' workers wait here
                int64_t a, b;
                range(i, a, b);
                r = run_pipeline(p, a, b, x, y, kern, selfs[i]);
            }
            rcs[i] = r;
            if (r) errs[i] = skdsp_last_error();
            char pb[256];
            skdsp_debug_path(pb, (int)sizeof(pb), 1);
            paths[i] = pb;
        });
    }
    {
        int64_t a, b;
        range(0, a, b);
        rcs[0] = run_pipeline(p, a, b, x, y, kern, selfs[0]);
    }
    for (auto &w : workers) w.join();
    for (int i = 1; i < nslots; ++i) {   // what the workers launched belongs to the caller's record: engine by engine, through the same de-duplication
        size_t at = 0;
        while (at < paths[i].size()) {
            size_t e = paths[i].find(',', at);
            if (e == std::string::npos) e = paths[i].size();
            if (e > at) note_path(paths[i].substr(at, e - at).c_str());
            at = e + 1;
        }
    }
    for (int i = 0; i < nslots; ++i)
        if (rcs[i]) {
            if (i > 0) set_error("%s", errs[i].c_str());
            return rcs[i];
        }
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
    if (!planar_only && !h->groups.empty() && !h->twin64 && !zi && !zf && opt().iir_par > 0) {
        // more than 8 biquads on a complex signal: group after group in place behind the first, each through the parallel form on the
        // interleaved samples where it applies (else whatever that group's own dispatch takes) -- not the whole cascade through two planes
        for (size_t gi = 0; gi < h->groups.size(); ++gi) {
            IirHandle *g = h->groups[gi];
            const void *src = gi == 0 ? x_dev : y_dev;
            int rc = iir_par_launch(g, src, n, 1, 0, 0, y_dev, s, 1, 1, 1);
            if (rc == 1) rc = iir_any_dev(g, src, n, y_dev);
            if (rc) return rc;
        }
        return SKDSP_OK;
    }
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
        // the parallel-form kernel zero-stuffs while it stages a segment: the L-fold signal is never written (1 = not applicable)
        if (L > 1 && opt().iir_par && opt().iir_up_fused && h->order == 2) {
            rc = iir_par_launch(h, x_dev, nl, 1, 0, 0, y_dev, s, 1, 0, L);
            if (rc != 1) return rc;
        }
        if ((rc = upsample_launch(x_dev, n, L, h->dtype, (double)L, y_dev, s))) return rc;
        return iir_any_dev(h, y_dev, nl, y_dev);
    }
    if (L > 1 && opt().iir_par && opt().iir_up_fused && h->order == 2 && !opt().iir_planar) {   // (interleaved in, interleaved out)
        rc = iir_par_launch(h, x_dev, nl, 1, 0, 0, y_dev, s, 1, 1, L);
        if (rc != 1) return rc;
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

static thread_local char g_path[256];
void skdsp::note_path(const char *engine)
{
    const size_t len = strlen(g_path), add = strlen(engine);
    if (len >= add && strcmp(g_path + len - add, engine) == 0 && (len == add || g_path[len - add - 1] == ',')) return;   // (the same engine again)
    if (add == 0) return;
    if (len + add + 2 >= sizeof(g_path)) return;
    if (len) g_path[len] = ',';
    memcpy(g_path + len + (len ? 1 : 0), engine, add + 1);
}

extern "C" {

const char *skdsp_last_error(void) { return g_err; }
int skdsp_debug_path(char *buf, int cap, int clear)
{
    if (buf && cap > 0) {
        strncpy(buf, g_path, (size_t)cap - 1);
        buf[cap - 1] = 0;
    }
    if (clear) g_path[0] = 0;
    return SKDSP_OK;
}
const char *skdsp_version(void) { return "skdsp-hip 0.1.0 (gfx950)"; }

int skdsp_init(int device)
{
    std::lock_guard<std::mutex> lk(ctx().mu);
    return init_locked(device);
}

static int shutdown_slot(Context &c)
{
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.ready) return SKDSP_OK;
    (void)hipSetDevice(c.device);
    (void)hipStreamSynchronize(c.stream);
    pipe_free(c);
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
        c.halo_flag = nullptr;
        (void)hipStreamDestroy(c.comm_stream);
        c.comm_stream = nullptr;
        c.ev_in = c.ev_halo = nullptr;
    }
    (void)hipStreamDestroy(c.stream);
    if (c.async_err) (void)hipHostFree(c.async_err);
    c.async_err = nullptr;
    c.ready = false;
    c.device = -1;
    return SKDSP_OK;
}

int skdsp_shutdown(void)
{
    for (int s = kMaxSlots - 1; s >= 0; --s) (void)shutdown_slot(ctx_of(s));
    std::lock_guard<std::mutex> lk(g_slots_mu);
    g_nslots = 0;
    return SKDSP_OK;
}

int skdsp_init_devices(const int *devices, int ndev)
{
    SK_CHECK(devices && ndev >= 1 && ndev <= kMaxSlots, SKDSP_ERR_BADARG, "init_devices: 1..%d devices", kMaxSlots);
    const int home = t_slot;
    int rc = SKDSP_OK;
    for (int s = 0; s < ndev && !rc; ++s) {
        t_slot = s;
        std::lock_guard<std::mutex> lk(ctx().mu);
        rc = init_locked(devices[s]);
    }
    t_slot = home;
    if (!rc && ctx().ready) SK_HIP(hipSetDevice(ctx().device));
    return rc;
}

int skdsp_slot_count(void) { return slot_count(); }

// the chunk planner of the host pipeline, exported for tests: chunk k of (n, L, M, hist) -> input / output ranges
int skdsp_host_chunk_plan(int64_t n, int L, int M, int64_t hist, int chunk_log2, int64_t k, int64_t *nchunks, int64_t *in_begin,
                          int64_t *in_end, int64_t *in_hist, int64_t *out_begin, int64_t *out_end)
{
    SK_CHECK(n >= 0 && L >= 1 && M >= 1 && hist >= 0, SKDSP_ERR_BADARG, "host_chunk_plan: bad arguments");
    const ChunkPlan p = plan_chunks(n, L, M, hist, 1, false, chunk_log2);
    if (nchunks) *nchunks = p.nchunks;
    SK_CHECK(k >= 0 && k < p.nchunks, SKDSP_ERR_BADARG, "host_chunk_plan: chunk %lld of %lld", (long long)k, (long long)p.nchunks);
    if (in_begin) *in_begin = p.in_begin(k);
    if (in_end) *in_end = p.in_end(k);
    if (in_hist) *in_hist = p.hist_of(k);
    if (out_begin) *out_begin = p.out_begin(k);
    if (out_end) *out_end = p.out_end(k);
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
        // (boxes without the marketing-name table -- /opt/amdgpu/share/libdrm/amdgpu.ids -- report an empty name: the architecture string then stands alone)
        if (prop.name[0]) snprintf(name, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
        else snprintf(name, (size_t)name_cap, "%s, %d CUs", prop.gcnArchName, prop.multiProcessorCount);
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
// page-locked host memory for result arrays (the Python layer recycles these blocks: _ffi.PinnedPool)
int skdsp_host_alloc(void **hptr, int64_t bytes)
{
    API_BEGIN;
    SK_CHECK(hptr && bytes > 0, SKDSP_ERR_BADARG, "host_alloc: bad arguments");
    SK_HIP(hipHostMalloc(hptr, (size_t)bytes, hipHostMallocPortable));
    return SKDSP_OK;
}
int skdsp_host_free(void *hptr)
{
    if (hptr) SK_HIP(hipHostFree(hptr));
    return SKDSP_OK;
}
int skdsp_memcpy_h2d(void *dst, const void *src, int64_t bytes)
{
    API_BEGIN;
    if (bytes > 0) SK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx().stream));
    return sync_checked();
}
int skdsp_memcpy_d2h(void *dst, const void *src, int64_t bytes)
{
    API_BEGIN;
    if (bytes > 0) SK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx().stream));
    return sync_checked();
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
    return sync_checked();
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
    ctx().last_timer_ms = (double)t;
    return async_err_check(ctx());
}
double skdsp_last_kernel_ms(void) { return ctx().ready ? ctx().last_timer_ms : -1.0; }
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
    return fir_updn_any(h, x_dev, n, n_hist, L, 1, y_dev);
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
    return fir_updn_any(h, x_dev, n, n_hist, L, M, y_dev);
}

// one chunk of a long host vector (run_pipeline): the same launch as the single-shot path, with the chunk's history
struct FirChunkJob {
    FirHandle *h;
    int mode, L, M;
};
static int fir_chunk_kernel(void *self, const void *x_dev, int64_t n_k, int64_t n_hist, void *y_dev, int64_t)
{
    const FirChunkJob *j = static_cast<const FirChunkJob *>(self);
    if (j->mode == 0) return fir_filter_any(j->h, x_dev, n_k, n_hist, y_dev);
    if (j->L == 1) return fir_dn_any(j->h, x_dev, n_k, n_hist, j->M, y_dev);
    return fir_updn_any(j->h, x_dev, n_k, n_hist, j->L, j->M, y_dev);
}
// the job on another slot: same filter, tables on that slot's device (clone made once, owned by the handle)
struct FirChunkJobs {
    FirChunkJob home;                 // the caller's handle
    FirChunkJob other[kMaxSlots];     // its clones, filled as slots ask for them
};
static void *fir_job_on_slot(void *base, int slot)
{
    FirChunkJobs *js = static_cast<FirChunkJobs *>(base);
    FirHandle *h = js->home.h;
    if (slot == h->slot) return &js->home;
    if ((int)h->clones.size() < kMaxSlots) h->clones.resize(kMaxSlots, nullptr);
    if (!h->clones[slot]) {
        h->clones[slot] = fir_derive(h, 0, h->ntaps, slot);
    }
    js->other[slot] = FirChunkJob{static_cast<FirHandle *>(h->clones[slot]), js->home.mode, js->home.L, js->home.M};
    return &js->other[slot];
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
    if (opt().host_pipeline && n > (((int64_t)3 << opt().host_chunk_log2) >> 1)) {
        // long vector: chunk pipeline (and every bound slot); exact by construction (n_hist)
        const int64_t hist = L > 1 ? (h->ntaps - 1 + L - 1) / L : h->ntaps - 1;
        const ChunkPlan p = plan_chunks(n, mode == 0 ? 1 : L, mode == 0 ? 1 : M, hist, esz, h->wide_out && !dtype_double(h->dtype),
                                        opt().host_chunk_log2);
        FirChunkJobs jobs;
        jobs.home = FirChunkJob{h, mode, L, M};
        return run_on_slots(p, (const char *)x, (char *)y, fir_chunk_kernel, fir_job_on_slot, &jobs, true);
    }
    void *x_dev = nullptr, *y_dev = nullptr;
    int rc = stage_in(x, (size_t)n * esz, &x_dev);
    if (rc) return rc;
    if ((rc = ws_reserve(1, (size_t)n_out * esz + 256, &y_dev))) return rc;
    if (mode == 0) rc = fir_filter_any(h, x_dev, n, 0, y_dev);
    else if (L == 1) rc = fir_dn_any(h, x_dev, n, 0, M, y_dev);
    else rc = fir_updn_any(h, x_dev, n, 0, L, M, y_dev);
    if (rc) return rc;
    return stage_out(y, y_dev, (size_t)n_out * esz, h);
}

// ---- N-D inputs: rows of one launch ---------------------------------------------------------------------------------
// lfilter(b, [1], x) filters along the last axis of an N-D array in one call (multirate_helper.py:108).  A FIR forgets
// after Ntaps-1 samples, so the rows are laid end to end with Ntaps-1 zeros between them and filtered as ONE signal from
// rest: every output sees exactly the window a launch over its row alone would see (zeros in front of every row), so the
// results agree to the kernels' rounding (the overlap-save tile boundaries fall elsewhere).  Host form: one pitched copy in, one launch, one pitched copy out;
// device form: the two pitched copies are device-to-device.  Costs (Ntaps-1)/n extra samples.
constexpr size_t kRowsBlockBudget = (size_t)16 << 30;   // bytes of the staged block of an N-D call (its output block is as large again)
static int64_t fir_rows_pitch(FirHandle *h, int64_t n) { return (int64_t)round_up((size_t)(n + fir_head(h, n)->ntaps - 1), 4); }

static int fir_rows_run(FirHandle *h, int64_t n, int64_t nrow, int64_t pitch, void *xp, void *yp)
{
    h = fir_head(h, n);   // (rows shorter than the filter: only the first n taps are ever reached)
    const size_t esz = dtype_size(h->dtype);
    // zeros between the rows (the last row needs none behind it)
    if (nrow > 1)
        SK_HIP(hipMemset2DAsync((char *)xp + (size_t)n * esz, (size_t)pitch * esz, 0, (size_t)(pitch - n) * esz, (size_t)(nrow - 1), ctx().stream));
    return fir_filter_any(h, xp, (nrow - 1) * pitch + n, 0, yp);
}

int skdsp_fir_filter_rows(skdsp_handle hh, const void *x, int64_t n, int64_t nrow, void *y)
{
    API_BEGIN;
    FirHandle *h = as_handle<FirHandle>(hh, H_FIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "fir_filter_rows: not a FIR handle");
    SK_CHECK(n >= 0 && nrow >= 0 && nrow < ((int64_t)1 << 31), SKDSP_ERR_BADARG, "fir_filter_rows: bad arguments");
    if (n == 0 || nrow == 0) return SKDSP_OK;
    SK_CHECK(x && y, SKDSP_ERR_BADARG, "fir_filter_rows: null buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t esz = dtype_size(h->dtype);
    const int64_t pitch = fir_rows_pitch(h, n);
    const size_t total = (size_t)nrow * (size_t)pitch * esz;
    // one block holds every row behind its zeros: refused beyond what a pitched copy takes / a sane staging budget (the caller then filters
    // row by row, each through the chunk pipeline)
    SK_CHECK((size_t)pitch * esz * 2 < ((size_t)1 << 31) && total <= kRowsBlockBudget, SKDSP_ERR_UNSUPPORTED,
             "fir_filter_rows: %lld rows of %lld samples do not fit one staged block (%zu bytes)", (long long)nrow, (long long)n, total);
    void *base = nullptr, *yp = nullptr;
    const bool wide = h->wide_out && !dtype_double(h->dtype);
    int rc = ws_reserve(0, kHeadroomBytes + (wide ? 2 : 1) * total + 256, &base);
    if (rc) return rc;
    void *xp = (char *)base + kHeadroomBytes;
    if ((rc = ws_reserve(1, total + 256, &yp))) return rc;
    SK_HIP(hipMemcpy2DAsync(xp, (size_t)pitch * esz, x, (size_t)n * esz, (size_t)n * esz, (size_t)nrow, hipMemcpyHostToDevice, ctx().stream));
    if ((rc = fir_rows_run(h, n, nrow, pitch, xp, yp))) return rc;
    size_t osz = esz;
    if (wide) {   // widen on the device (the staged input is done with in stream order)
        if ((rc = widen_launch(yp, (int64_t)(total / 4), xp, ctx().stream))) return rc;
        yp = xp;
        osz = 2 * esz;
    }
    SK_HIP(hipMemcpy2DAsync(y, (size_t)n * osz, yp, (size_t)pitch * osz, (size_t)n * osz, (size_t)nrow, hipMemcpyDeviceToHost, ctx().stream));
    return sync_checked();
}

int skdsp_fir_filter_rows_dev(skdsp_handle hh, const void *x_dev, int64_t n, int64_t nrow, int64_t x_stride, int64_t y_stride, void *y_dev)
{
    API_BEGIN;
    FirHandle *h = as_handle<FirHandle>(hh, H_FIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "fir_filter_rows: not a FIR handle");
    SK_CHECK(n >= 0 && nrow >= 0 && nrow < ((int64_t)1 << 31), SKDSP_ERR_BADARG, "fir_filter_rows: bad arguments");
    if (n == 0 || nrow == 0) return SKDSP_OK;
    SK_CHECK(x_stride >= n && y_stride >= n, SKDSP_ERR_BADARG, "fir_filter_rows: row stride below the row length");
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t esz = dtype_size(h->dtype);
    const int64_t pitch = fir_rows_pitch(h, n);
    const size_t total = (size_t)nrow * (size_t)pitch * esz;
    SK_CHECK((size_t)pitch * esz < ((size_t)1 << 31) && (size_t)x_stride * esz < ((size_t)1 << 31) && (size_t)y_stride * esz < ((size_t)1 << 31) &&
                 total <= kRowsBlockBudget, SKDSP_ERR_UNSUPPORTED,
             "fir_filter_rows_dev: %lld rows of %lld samples do not fit one staged block (%zu bytes)", (long long)nrow, (long long)n, total);
    void *base = nullptr, *yp = nullptr;
    int rc = ws_reserve(0, kHeadroomBytes + total + 256, &base);
    if (rc) return rc;
    void *xp = (char *)base + kHeadroomBytes;
    if ((rc = ws_reserve(1, total + 256, &yp))) return rc;
    SK_HIP(hipMemcpy2DAsync(xp, (size_t)pitch * esz, x_dev, (size_t)x_stride * esz, (size_t)n * esz, (size_t)nrow, hipMemcpyDeviceToDevice, ctx().stream));
    if ((rc = fir_rows_run(h, n, nrow, pitch, xp, yp))) return rc;
    SK_HIP(hipMemcpy2DAsync(y_dev, (size_t)y_stride * esz, yp, (size_t)pitch * esz, (size_t)n * esz, (size_t)nrow, hipMemcpyDeviceToDevice, ctx().stream));
    return SKDSP_OK;
}

int skdsp_fir_filter(skdsp_handle h, const void *x, int64_t n, void *y) { return fir_host_call(h, x, n, 1, 1, 0, y); }

int skdsp_fir_filter_sharded(skdsp_handle h, const void *x, int64_t n, void *y, int ngpu)
{
    {
        API_BEGIN;
        SK_CHECK(ngpu >= 0 && ngpu <= slot_count(), SKDSP_ERR_BADARG, "fir_filter_sharded: ngpu = %d, %d slots bound (skdsp_init_devices)", ngpu,
                 slot_count());
    }
    tl_slot_limit = ngpu;
    const int rc = fir_host_call(h, x, n, 1, 1, 0, y);
    tl_slot_limit = 0;
    return rc;
}
int skdsp_fir_up(skdsp_handle h, const void *x, int64_t n, int L, void *y) { return fir_host_call(h, x, n, L, 1, 1, y); }
int skdsp_fir_dn(skdsp_handle h, const void *x, int64_t n, int M, void *y) { return fir_host_call(h, x, n, 1, M, 1, y); }
int skdsp_fir_updn(skdsp_handle h, const void *x, int64_t n, int L, int M, void *y) { return fir_host_call(h, x, n, L, M, 1, y); }

// ------------------------------------------------------------------------ IIR
// seq_limit > 0: the spread above which the handle runs the reference's recursion, given by the caller instead of derived from `dtype` -- the
// float64 twin of a float32 handle serves the float32 contract, so it is probed against the float32 limit its parent just passed (with the
// float64 limit a cheby1(26) cascade, spread 3.7e-11, would have run sample by sample although 1e-6 never needed it)
static int iir_create_common(int nsec, int order, const std::vector<double> &coef, int dtype, skdsp_handle *out, double seq_limit = 0.0)
{
    SK_CHECK(out, SKDSP_ERR_BADARG, "iir_create: null out");
    SK_CHECK(dtype_valid(dtype), SKDSP_ERR_BADARG, "iir_create: bad dtype %d", dtype);
    SK_CHECK(order == 2 && nsec >= 1 && nsec <= 4096, SKDSP_ERR_UNSUPPORTED,
             "iir_create: %d sections of order %d not supported (second-order sections, 1 .. 4096 of them)", nsec, order);
    std::unique_ptr<IirHandle> h(new IirHandle());
    h->kind = H_IIR;
    h->dtype = dtype;
    h->nsec = nsec;
    h->order = order;
    h->coef = coef;
    if (nsec > 8 && opt().iir_seq != 0) {
        // How far apart do two float64 evaluations of THIS cascade lie -- the reference's recursion with the sections as given and in reverse
        // order (equal in exact arithmetic)?  A 40th-order Chebyshev design shows 1e-7 .. 1e-6 of its output; the scans (which combine chunk
        // transitions instead of running the recursion) add 30 - 400 x that on such cascades (profiles/r05/iir_illcond.txt), which would carry
        // them past the contract (1e-6 of the output for float32 signals, 1e-10 for float64 ones).  Such a handle runs the recursion itself
        // (iir_seq.hip): slow, and bit for bit the reference's result.  Cascades of up to 8 sections are not probed: the parallel form's own
        // acceptance test covers them.
        const int NH = 4096;
        std::vector<double> u(NH), v(NH);
        unsigned long long lcg = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < NH; ++i) {   // (sum of four uniforms: bell-shaped, unit-level, reproducible)
            double a = 0.0;
            for (int k = 0; k < 4; ++k) {
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                a += (double)(lcg >> 11) / 9007199254740992.0 - 0.5;
            }
            u[i] = v[i] = a * 1.7320508075688772;
        }
        auto run = [&](std::vector<double> &w, bool rev) {
            for (int k = 0; k < nsec; ++k) {
                const double *c = coef.data() + 5 * (rev ? nsec - 1 - k : k);
                double z0 = 0.0, z1 = 0.0;
                for (int i = 0; i < NH; ++i) {
                    const double xn = w[i], xc = c[0] * xn + z0;
                    z0 = c[1] * xn - c[3] * xc + z1;
                    z1 = c[2] * xn - c[4] * xc;
                    w[i] = xc;
                }
            }
        };
        run(u, false);
        run(v, true);
        double peak = 0.0, diff = 0.0;
        for (int i = 0; i < NH; ++i) {
            peak = std::max(peak, std::fabs(u[i]));
            diff = std::max(diff, std::fabs(u[i] - v[i]));
        }
        const double spread = std::isfinite(diff) && peak > 0.0 ? diff / peak : 1.0;
        h->seq_spread = spread;
        const double limit = seq_limit > 0.0 ? seq_limit : dtype_double(dtype) ? 2.5e-13 : 2.5e-9;   // (400 x spread stays inside the contract)
        if (opt().iir_seq == 2 || !(spread <= limit)) {
            h->seq = true;
            h->seq_coef = coef;
            *out = h.release();
            return SKDSP_OK;
        }
    }
    if (nsec > 8) {
        // groups of at most 8 sections, as even as possible (10 -> 5 + 5): each a handle of its own, made from the CALLER's factorisation.
        // Between two groups the signal is stored in the handle's precision.  For float32 handles that rounding (6e-8 of the
        // INTERMEDIATE's peak, then amplified by the rest of the cascade) must stay below the float32 contract on the output: with
        // A = l1 norm of the impulse response up to a boundary, B = from it on, T = of the whole cascade, the boundary costs at most
        // 6e-8 A B / T of the output's scale.  A Butterworth cascade has A B / T ~ 2; an order-17 Chebyshev in scipy's section order
        // 170 (measured: 1e-5).  Such cascades keep the largest groups the cascade kernels take (12: no boundary at all up to 12 sections).
        int per = 8;
        if (!dtype_double(dtype)) {
            const int NH = 16384;
            auto run = [&](int s0, int s1, std::vector<double> &v) {   // v <- sections [s0, s1) applied to v (DF2T, from rest)
                for (int sct = s0; sct < s1; ++sct) {
                    const double *c = coef.data() + 5 * sct;
                    double z0 = 0.0, z1 = 0.0;
                    for (int i = 0; i < NH; ++i) {
                        const double xin = v[i], yo = c[0] * xin + z0;
                        z0 = c[1] * xin - c[3] * yo + z1;
                        z1 = c[2] * xin - c[4] * yo;
                        v[i] = yo;
                    }
                }
            };
            auto l1 = [&](const std::vector<double> &v) { double a = 0.0; for (double q : v) a += std::fabs(q); return a; };
            std::vector<double> imp(NH, 0.0);
            imp[0] = 1.0;
            const int ng8 = (nsec + 7) / 8;
            // the boundaries' costs ADD UP (ng8 - 1 of them), so their sum is what is bounded; the l1 norms behind every boundary come from
            // ONE backward pass (sections commute: the tail from boundary g is the sections of group g applied to the tail from boundary
            // g + 1), the ones in front of it from one forward pass: O(nsec NH) in all.  Cascades of more than 256 sections are not
            // analysed (seconds of host work inside handle creation): they take the float64 twin.
            double worst = 1e300;
            if (nsec <= 256) {
                std::vector<int> first(ng8 + 1, 0);
                for (int g = 0; g < ng8; ++g) first[g + 1] = first[g] + nsec / ng8 + (g < nsec % ng8 ? 1 : 0);
                std::vector<double> tail_l1(ng8 + 1, 0.0), v = imp;
                for (int g = ng8 - 1; g >= 1; --g) {   // v = impulse response of the sections [first[g], nsec)
                    run(first[g], first[g + 1], v);
                    tail_l1[g] = l1(v);
                }
                run(first[0], first[1], v);
                const double T = l1(v);
                std::vector<double> head = imp;
                worst = 0.0;
                for (int g = 0; g + 1 < ng8; ++g) {
                    run(first[g], first[g + 1], head);
                    worst += l1(head) * tail_l1[g + 1] / std::max(T, 1e-300);
                }
            }
            if (!(worst <= 16.0)) {
                if (nsec <= 12) goto single_group;      // one launch sequence of the cascade kernels: float64 between ALL sections
                // too many sections for that, and no float32 boundary is safe: the same cascade in float64 (widen, filter, narrow)
                skdsp_handle th = nullptr;
                const int rc = iir_create_common(nsec, 2, coef, dtype == SKDSP_C64 ? SKDSP_C128 : SKDSP_F64, &th, 2.5e-9);
                if (rc) return rc;
                h->twin64 = static_cast<IirHandle *>(th);
                h->twin64->slot = ctx().slot;
                *out = h.release();
                return SKDSP_OK;
            }
        }
        const int ng = (nsec + per - 1) / per;
        for (int g = 0, s0 = 0; g < ng; ++g) {
            const int cnt = nsec / ng + (g < nsec % ng ? 1 : 0);
            std::vector<double> part(coef.begin() + (size_t)5 * s0, coef.begin() + (size_t)5 * (s0 + cnt));
            skdsp_handle gh = nullptr;
            const int rc = iir_create_common(cnt, 2, part, dtype, &gh);
            if (rc) return rc;
            IirHandle *gp = static_cast<IirHandle *>(gh);
            gp->group_first = s0;
            gp->slot = ctx().slot;
            h->groups.push_back(gp);
            s0 += cnt;
        }
        *out = h.release();
        return SKDSP_OK;
    }
single_group:
    if (order == 2 && nsec >= 2) {
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

int skdsp_iir_sequential(skdsp_handle hh, int *is_sequential, double *spread)
{
    HandleBase *hb = static_cast<HandleBase *>(hh);
    SK_CHECK(hb && hb->kind == H_IIR, SKDSP_ERR_BADARG, "iir_sequential: not an IIR handle");
    IirHandle *h = static_cast<IirHandle *>(hb);
    if (is_sequential) *is_sequential = (h->seq || (h->twin64 && h->twin64->seq)) ? 1 : 0;   // (what actually runs: a float32 handle may filter through its float64 twin)
    if (spread) *spread = h->seq_spread;
    return SKDSP_OK;
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

int skdsp_sos_par_info(const double *sos, int nsec, double *out, int *accepted)
{
    SK_CHECK(sos && out && accepted, SKDSP_ERR_BADARG, "sos_par_info: null argument");
    SK_CHECK(nsec >= 1 && nsec <= 8, SKDSP_ERR_UNSUPPORTED, "sos_par_info: 1..8 biquads");
    std::vector<double> coef((size_t)nsec * 5);
    for (int s = 0; s < nsec; ++s) {
        const double *q = sos + 6 * s;
        SK_CHECK(q[3] == 1.0, SKDSP_ERR_BADARG, "sos[:, 3] should be all ones");
        double *c = coef.data() + 5 * s;
        c[0] = q[0]; c[1] = q[1]; c[2] = q[2]; c[3] = q[4]; c[4] = q[5];
    }
    return iir_par_expand_host(coef.data(), nsec, out, accepted);
}

// rows of one launch (parallel form) or, where that does not apply, row by row through the cascade kernels
static int iir_rows_dev(IirHandle *h, const void *x_dev, int64_t n, int64_t nrow, int64_t x_stride, int64_t y_stride, void *y_dev)
{
    if (n <= 0 || nrow <= 0) return SKDSP_OK;
    SK_CHECK(x_stride >= n && y_stride >= n, SKDSP_ERR_BADARG, "iir_filter_rows: row stride below the row length");
    SK_CHECK(nrow < (1 << 24), SKDSP_ERR_BADARG, "iir_filter_rows: too many rows");
    const size_t esz = dtype_size(h->dtype);
    // the reference's recursion takes all rows in ONE launch (one wave per row), not one single-wave kernel per row in stream order
    if (h->seq && !dtype_complex(h->dtype)) return iir_seq_launch(h, x_dev, n, (int)nrow, x_stride, y_stride, y_dev, ctx().stream);
    if (!h->groups.empty() && !dtype_complex(h->dtype) && opt().iir_par > 0) {
        // groups of sections (more than 8 biquads): every group over all rows in one launch where its parallel form applies, in place behind the first
        for (size_t gi = 0; gi < h->groups.size(); ++gi) {
            IirHandle *g = h->groups[gi];
            const void *src = gi == 0 ? x_dev : y_dev;
            const int64_t ss = gi == 0 ? x_stride : y_stride;
            int rc = iir_par_launch(g, src, n, (int)nrow, ss, y_stride, y_dev, ctx().stream);
            if (rc == 1) {
                for (int64_t r = 0; r < nrow; ++r)
                    if ((rc = iir_any_dev(g, (const char *)src + (size_t)r * ss * esz, n, (char *)y_dev + (size_t)r * y_stride * esz))) return rc;
            } else if (rc) {
                return rc;
            }
        }
        return SKDSP_OK;
    }
    if (!dtype_complex(h->dtype) && opt().iir_par > 0) {
        const int r = iir_par_launch(h, x_dev, n, (int)nrow, x_stride, y_stride, y_dev, ctx().stream);
        if (r != 1) return r;
    }
    for (int64_t r = 0; r < nrow; ++r) {
        const int rc = iir_any_dev(h, (const char *)x_dev + (size_t)r * x_stride * esz, n, (char *)y_dev + (size_t)r * y_stride * esz);
        if (rc) return rc;
    }
    return SKDSP_OK;
}

int skdsp_iir_filter_rows_dev(skdsp_handle hh, const void *x_dev, int64_t n, int64_t nrow, int64_t x_stride, int64_t y_stride, void *y_dev)
{
    API_BEGIN;
    IirHandle *h = as_handle<IirHandle>(hh, H_IIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "iir_filter_rows: not an IIR handle");
    std::lock_guard<std::mutex> lk(h->mu);
    return iir_rows_dev(h, x_dev, n, nrow, x_stride, y_stride, y_dev);
}

int skdsp_iir_filter_rows(skdsp_handle hh, const void *x, int64_t n, int64_t nrow, void *y)
{
    API_BEGIN;
    IirHandle *h = as_handle<IirHandle>(hh, H_IIR);
    SK_CHECK(h, SKDSP_ERR_BADARG, "iir_filter_rows: not an IIR handle");
    SK_CHECK(n >= 0 && nrow >= 0, SKDSP_ERR_BADARG, "iir_filter_rows: bad arguments");
    if (n == 0 || nrow == 0) return SKDSP_OK;
    SK_CHECK(x && y, SKDSP_ERR_BADARG, "iir_filter_rows: null buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t esz = dtype_size(h->dtype), bytes = (size_t)n * (size_t)nrow * esz;
    void *x_dev = nullptr, *y_dev = nullptr;
    int rc = stage_in(x, bytes, &x_dev);
    if (rc) return rc;
    if ((rc = ws_reserve(1, bytes + 256, &y_dev))) return rc;
    if ((rc = iir_rows_dev(h, x_dev, n, nrow, n, n, y_dev))) return rc;
    return stage_out(y, y_dev, bytes, h);
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

struct IirChunkJob {
    IirHandle *h;
    std::vector<double> state;  // state after the previous chunk, in the caller-visible (scipy zi) convention
};
static int iir_chunk_kernel(void *self, const void *x_dev, int64_t n_k, int64_t, void *y_dev, int64_t k)
{
    IirChunkJob *j = static_cast<IirChunkJob *>(self);
    std::vector<double> zf(j->state.size());
    int rc = iir_any_dev(j->h, x_dev, n_k, y_dev, k == 0 ? nullptr : j->state.data(), zf.data());
    j->state.swap(zf);
    return rc;
}
static void *iir_job_on_slot(void *base, int) { return base; }

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
    if (L == 1 && M == 1 && opt().host_pipeline && n > (((int64_t)3 << opt().host_chunk_log2) >> 1)) {
        // long vector: chunk pipeline on the caller's slot, the recursion carried from chunk to chunk as zi / zf
        const ChunkPlan p = plan_chunks(n, 1, 1, 0, esz, h->wide_out && !dtype_double(h->dtype), opt().host_chunk_log2);
        IirChunkJob job;
        job.h = h;
        job.state.assign((size_t)(dtype_complex(h->dtype) ? 2 : 1) * h->nsec * h->order, 0.0);
        return run_on_slots(p, (const char *)x, (char *)y, iir_chunk_kernel, iir_job_on_slot, &job, false);
    }
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
// (SURVEY.md 8(b)'s names for the same three calls)
int skdsp_sos_filter(skdsp_handle h, const void *x, int64_t n, void *y) { return skdsp_iir_filter(h, x, n, y); }
int skdsp_sos_up(skdsp_handle h, const void *x, int64_t n, int L, void *y) { return skdsp_iir_up(h, x, n, L, y); }
int skdsp_sos_dn(skdsp_handle h, const void *x, int64_t n, int M, void *y) { return skdsp_iir_dn(h, x, n, M, y); }

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
