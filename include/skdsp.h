/*
 * skdsp.h -- C ABI of libskdsp_hip.so: the MI355X (gfx950) replacement for the
 * native kernels that scikit-dsp-comm's streaming-filter hot path executes.
 *
 * The reference has NO FFI of its own for this path: it is pure Python whose
 * arithmetic is one-line calls into scipy.signal / numpy.  The entry points
 * below are therefore "what a binding for this path would bind": one entry per
 * reference call site, cited as /root/reference/src/sk_dsp_comm/<file>:<line>.
 * INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success or a negative skdsp_status; the
 *     message for the calling thread is in skdsp_last_error().
 *   - dtype = arithmetic/storage type of the signal: F32/C64 (hot path),
 *     F64/C128 (direct-form kernels in double, for float64 callers).
 *     Complex is interleaved (re,im), C-contiguous, as NumPy stores it.
 *   - coefficient arrays are always host float64 / complex128 (what the
 *     reference objects hold); the library rounds once to the signal dtype.
 *   - "*_dev" variants take DEVICE pointers (from skdsp_malloc) and run
 *     asynchronously on the library stream; host variants copy in/out and
 *     return after the stream is idle.  The caller owns every buffer; the
 *     library copies coefficients at create() and never retains x/y.
 *   - n_hist: number of valid samples stored immediately BEFORE x_dev[0] in the
 *     same allocation (x_dev[-n_hist..-1]).  0 = zero initial state, which is
 *     what every reference call uses (lfilter/sosfilt without zi).  The sharded
 *     path uses it for the Ntaps-1 halo received from the left neighbour.
 *   - a handle serialises its own calls.  Calls lock the SLOT they run on (a slot = one GPU binding with its stream
 *     and workspaces): caller threads all use slot 0, so their calls take turns on its stream; the worker threads of
 *     a multi-slot host call run concurrently, one per slot.  ctypes releases the GIL around each call.
 */
#ifndef SKDSP_H
#define SKDSP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { SKDSP_F32 = 0, SKDSP_C64 = 1, SKDSP_F64 = 2, SKDSP_C128 = 3 } skdsp_dtype;

typedef enum {
    SKDSP_OK = 0,
    SKDSP_ERR_BADARG = -1,   /* -> ValueError / TypeError in the Python mirror */
    SKDSP_ERR_NODEVICE = -2, /* no HIP device: the product path fails loudly   */
    SKDSP_ERR_NOMEM = -3,
    SKDSP_ERR_HIP = -4,
    SKDSP_ERR_RCCL = -5,
    SKDSP_ERR_UNSUPPORTED = -6
} skdsp_status;

/* FIR algorithm selector (skdsp_fir_set_algo). AUTO picks by dtype / tap count. */
typedef enum { SKDSP_FIR_AUTO = 0, SKDSP_FIR_DIRECT = 1, SKDSP_FIR_OLS = 2 } skdsp_fir_algo;

typedef void *skdsp_handle;

/* ---- runtime ------------------------------------------------------------ */
int skdsp_init(int device);               /* bind the caller's slot (slot 0) to one GPU, create its stream */
/* Bind slots 0 .. ndev-1 to the listed GPUs (slot 0 = the device every *_dev entry point and handle uses).  With more
 * than one slot bound, the host-pointer FIR entry points deal the chunks of a long vector to all of them, one worker
 * thread, stream and PCIe link each (the reference's single call multirate_FIR(b).filter(x), multirate_helper.py:104-109,
 * then scales over the node without a launcher).  A device may be listed more than once. */
int skdsp_init_devices(const int *devices, int ndev);
int skdsp_slot_count(void);
int skdsp_shutdown(void);
int skdsp_device_count(void);
int skdsp_device_info(char *name, int name_cap, int *compute_units, int64_t *hbm_bytes, int *clock_khz);
const char *skdsp_last_error(void);
/* Test / diagnostic aid (no reference counterpart): the comma-joined names of the kernels families the calling thread's API calls have
 * launched since the last clear ("fir_ols", "fir_bx", "fir_up4k", "iir_par", ...), so that a parity test can assert that a shape
 * reaches the engine it means to exercise.  clear != 0 empties the record. */
int skdsp_debug_path(char *buf, int cap, int clear);
const char *skdsp_version(void);
/* Run-time switches (algorithm A/B selectors, pipeline chunk size, ...).  Each option NAME is read once from the
 * environment variable SKDSP_<NAME> when the library first needs it; afterwards only these calls change it, so no
 * launch path calls getenv().  Names: see struct Options in csrc/skdsp_internal.hpp.  Unknown name -> BADARG. */
int skdsp_set_option(const char *name, int value);
int skdsp_get_option(const char *name, int *value);

int skdsp_malloc(void **dptr, int64_t bytes);
int skdsp_free(void *dptr);
/* Page-locked host memory.  A copy back into a FRESH pageable array (np.empty) runs at ~25-30 GB/s on this platform
 * (page population + first device access), into page-locked memory at the link rate (56 GB/s): the Python layer hands
 * out result arrays backed by recycled blocks from here. */
int skdsp_host_alloc(void **hptr, int64_t bytes);
int skdsp_host_free(void *hptr);
int skdsp_memcpy_h2d(void *dst_dev, const void *src_host, int64_t bytes);
int skdsp_memcpy_d2h(void *dst_host, const void *src_dev, int64_t bytes);
int skdsp_memcpy_d2d(void *dst_dev, const void *src_dev, int64_t bytes);
int skdsp_memset(void *dst_dev, int value, int64_t bytes);
int skdsp_sync(void);                      /* wait for the library stream */
/* HIP-event stopwatch on the library stream (bench.py times kernels with it) */
int skdsp_timer_start(void);
int skdsp_timer_stop(float *elapsed_ms);   /* synchronises on the stop event */
/* what the last skdsp_timer_stop of the calling thread's slot measured, in ms (SURVEY.md 8(b)'s name for it); -1 before the first */
double skdsp_last_kernel_ms(void);
/* fill a device buffer with counter-based N(0,1)/sqrt(2) complex (or N(0,1) real)
 * noise keyed by (seed, global sample index): any window can be regenerated. */
int skdsp_fill_noise_dev(void *x_dev, int64_t n, int dtype, uint64_t seed, int64_t first_index);

/* ---- FIR: multirate_FIR (multirate_helper.py:95-127) ---------------------- */
/* ctor, multirate_helper.py:95-101.  taps: ntaps float64 (taps_complex=0) or
 * ntaps complex128 (taps_complex=1). */
int skdsp_fir_create(const void *taps, int ntaps, int taps_complex, int dtype, skdsp_handle *out);
int skdsp_fir_set_algo(skdsp_handle h, int algo);
int skdsp_fir_get_algo(skdsp_handle h, int64_t n, int *algo_used);
/* .filter(x): signal.lfilter(b,[1],x), multirate_helper.py:104-109.  y has n samples. */
int skdsp_fir_filter(skdsp_handle h, const void *x, int64_t n, void *y);
int skdsp_fir_filter_dev(skdsp_handle h, const void *x_dev, int64_t n, int64_t n_hist, void *y_dev);
/* The same call spread over the GPUs of one process (SURVEY 8(b) / BASELINE config 5 without a launcher): the host vector is cut into
 * contiguous sample blocks, one range of chunks per slot bound by skdsp_init_devices, each chunk staged with the Ntaps-1 samples in
 * front of it -- the halo comes from the caller's array, so the GPUs exchange nothing.  ngpu: slots to use, 0 = all bound
 * (skdsp_fir_filter uses all of them as well; this entry lets a caller choose).  Device-resident shards with an RCCL halo
 * between processes: skdsp_fir_filter_shard_dev below. */
int skdsp_fir_filter_sharded(skdsp_handle h, const void *x, int64_t n, void *y, int ngpu);
/* N-D inputs: lfilter filters along the last axis in ONE call (multirate_helper.py:108).  nrow rows of n samples, each
 * filtered from rest: one pitched copy in, one launch over the rows laid end to end with Ntaps-1 zeros between them,
 * one pitched copy out (host form: rows contiguous; device form: x_stride / y_stride elements between rows, >= n). */
int skdsp_fir_filter_rows(skdsp_handle h, const void *x, int64_t n, int64_t nrow, void *y);
int skdsp_fir_filter_rows_dev(skdsp_handle h, const void *x_dev, int64_t n, int64_t nrow, int64_t x_stride, int64_t y_stride, void *y_dev);
/* .up(x,L): lfilter(b,[1], L*upsample(x,L)), multirate_helper.py:112-118.  y has n*L samples. */
int skdsp_fir_up(skdsp_handle h, const void *x, int64_t n, int L, void *y);
int skdsp_fir_up_dev(skdsp_handle h, const void *x_dev, int64_t n, int64_t n_hist, int L, void *y_dev);
/* .dn(x,M): downsample(lfilter(b,[1],x), M), multirate_helper.py:121-127.  y has n/M samples. */
int skdsp_fir_dn(skdsp_handle h, const void *x, int64_t n, int M, void *y);
int skdsp_fir_dn_dev(skdsp_handle h, const void *x_dev, int64_t n, int64_t n_hist, int M, void *y_dev);
/* fused rational resampler = downsample(.up(x,L), M) (BASELINE.json config 3;
 * sigsys.py:3078-3083 applied to multirate_helper.py:112-118).  y has (n*L)/M samples. */
int skdsp_fir_updn(skdsp_handle h, const void *x, int64_t n, int L, int M, void *y);
int skdsp_fir_updn_dev(skdsp_handle h, const void *x_dev, int64_t n, int64_t n_hist, int L, int M, void *y_dev);

/* Is this cascade one that runs the reference's own recursion, sample by sample (csrc/iir_seq.hip)?  scipy.signal.sosfilt
 * (multirate_helper.py:173) evaluates a cascade section after section in float64; for ill-conditioned designs (a 40th-order Chebyshev)
 * that result is itself good to ~1e-6 only, and a scan would add to it.  skdsp_sos_create probes every cascade of more than 8 sections
 * (the reference's recursion with the sections as given and reversed, on a fixed pseudo-random input); *spread receives the relative
 * difference it found (0 where no probe ran), *is_sequential whether the handle therefore runs sequentially (slow: the reference's own
 * speed).  The Python layer logs a WARNING for such handles. */
int skdsp_iir_sequential(skdsp_handle h, int *is_sequential, double *spread);

/* ---- IIR: multirate_IIR (multirate_helper.py:159-192), rate_change (:54-83) - */
/* sos: nsec x 6 float64, sos[:,3]==1 (scipy.signal.sosfilt contract); any number of sections up to 4096, as sosfilt takes them
 * (more than 8 run as consecutive groups of at most 8 on the device; a float32 cascade whose intermediate signals do not survive
 * float32 storage between two groups runs in float64 inside). */
int skdsp_sos_create(const double *sos, int nsec, int dtype, skdsp_handle *out);
/* transfer function (b,a) for signal.lfilter(b,a,.), multirate_helper.py:74,81;
 * a[0]-normalised.  scipy runs (b,a) as one DF2T section of order N; the scan kernel
 * runs the same transfer function factored into biquads (see capi.hip: the order-N
 * companion coordinates are too ill-conditioned for an affine scan).  Order <= 24. */
int skdsp_tf_create(const double *b, int nb, const double *a, int na, int dtype, skdsp_handle *out);
/* host-only: the (b,a) -> sos factorisation tf_create applies; sos_out holds up to 12x6 doubles */
int skdsp_tf2sos(const double *b, int nb, const double *a, int na, double *sos_out, int *nsec_out);
/* .filter: sosfilt(sos,x) (:169-174) / lfilter(b,a,x) */
int skdsp_iir_filter(skdsp_handle h, const void *x, int64_t n, void *y);
int skdsp_iir_filter_dev(skdsp_handle h, const void *x_dev, int64_t n, void *y_dev);
/* Streaming state (NOT reference behaviour: the reference always starts from
 * rest).  zi / zf are host arrays of skdsp_iir_state_len() doubles: for a real signal the DF2T
 * delays in scipy.signal.sosfilt's zi layout (n_sections x 2); for a complex signal the states of
 * the real-part stream followed by those of the imaginary-part stream.  Handles made by
 * skdsp_tf_create use the layout of their internal biquad cascade (opaque: feed a zf back as zi).
 * zi == NULL means rest; zf == NULL means "not wanted" (no stream synchronisation). */
int skdsp_iir_state_len(skdsp_handle h, int *len);
int skdsp_iir_filter_state_dev(skdsp_handle h, const void *x_dev, int64_t n, const double *zi, double *zf, void *y_dev);
/* N-D inputs: scipy filters along the last axis in ONE call (multirate_helper.py:173: sosfilt(sos, x), x of any
 * shape).  nrow rows of n samples, x_stride / y_stride elements apart (>= n), zero initial state per row, one launch
 * (one staged copy each way for the host-pointer form, rows contiguous).  */
int skdsp_iir_filter_rows(skdsp_handle h, const void *x, int64_t n, int64_t nrow, void *y);
int skdsp_iir_filter_rows_dev(skdsp_handle h, const void *x_dev, int64_t n, int64_t nrow, int64_t x_stride, int64_t y_stride, void *y_dev);
/* host-only (no GPU): the partial-fraction expansion the parallel-form scan (csrc/iir_par.hip) runs for the transfer
 * function of an (n_sections x 6) sos array, H(z) = c0 + sum_k (r0_k + r1_k z^-1) / (1 + a1_k z^-1 + a2_k z^-2):
 * out = [c0, (a1, a2, r0, r1) x n_sections, branch-cancellation factor, impulse-response error vs the cascade,
 *        worst probe error of the float32 from-rest states on 128-sample chunks, the same on 96-sample chunks] (5 + 4 n_sections doubles:
 *        float32 / complex64 signals through 7 - 8 biquads form those states on the float32 matrix instruction where the error stays below 5e-7);
 * *accepted = 1 when the expansion passed its acceptance test (else the cascade kernels serve the handle). */
int skdsp_sos_par_info(const double *sos, int nsec, double *out, int *accepted);
/* .up: filter(L*upsample(x,L)) (:69-75, :177-183); y has n*L samples */
int skdsp_iir_up(skdsp_handle h, const void *x, int64_t n, int L, void *y);
int skdsp_iir_up_dev(skdsp_handle h, const void *x_dev, int64_t n, int L, void *y_dev);
/* .dn: downsample(filter(x), M) (:77-83, :186-192); y has n/M samples */
int skdsp_iir_dn(skdsp_handle h, const void *x, int64_t n, int M, void *y);
int skdsp_iir_dn_dev(skdsp_handle h, const void *x_dev, int64_t n, int M, void *y_dev);
/* The names SURVEY.md 8(b) sketched for the three calls above on handles made by skdsp_sos_create: the same functions
 * (multirate_IIR.filter / .up / .dn, multirate_helper.py:169-192), kept so that a binding written from that table links. */
int skdsp_sos_filter(skdsp_handle h, const void *x, int64_t n, void *y);
int skdsp_sos_up(skdsp_handle h, const void *x, int64_t n, int L, void *y);
int skdsp_sos_dn(skdsp_handle h, const void *x, int64_t n, int M, void *y);

/* ---- rate-change primitives (sigsys.py:3031-3083) -------------------------- */
/* upsample: y[k*L] = x[k], zeros elsewhere (sigsys.py:3050-3053). y has n*L samples. */
int skdsp_upsample(const void *x, int64_t n, int L, int dtype, void *y);
int skdsp_upsample_dev(const void *x_dev, int64_t n, int L, int dtype, double scale, void *y_dev);
/* downsample: y[k] = x[k*M+p], k < n/M, 0 <= p < M (sigsys.py:3078-3083). */
int skdsp_downsample(const void *x, int64_t n, int M, int p, int dtype, void *y);
int skdsp_downsample_dev(const void *x_dev, int64_t n, int M, int p, int dtype, void *y_dev);

/* Host-pointer entry points of a float32/complex64 handle deliver y as float64/complex128 (the
 * reference's result dtype, multirate_helper.py:108 etc.): widened on the device before the copy
 * back, so y must hold twice the bytes.  Device-pointer (_dev) entry points are not affected. */
/* The chunk planner of the host-pointer entry points (long vectors are pipelined in chunks of 2^chunk_log2 samples that
 * are exact continuations of each other): ranges of chunk k.  Host-only, for tests and for callers who stream themselves. */
int skdsp_host_chunk_plan(int64_t n, int L, int M, int64_t hist, int chunk_log2, int64_t k, int64_t *nchunks, int64_t *in_begin,
                          int64_t *in_end, int64_t *in_hist, int64_t *out_begin, int64_t *out_end);
int skdsp_set_wide_output(skdsp_handle h, int on);
int skdsp_destroy(skdsp_handle h);

/* ---- sample-block sharding across GPUs (one process per GPU, RCCL) ------- */
/* rank 0 creates the 128-byte RCCL unique id; the launcher hands it to every rank. */
int skdsp_dist_unique_id(void *id128);
int skdsp_dist_init(int rank, int world, const void *id128);
int skdsp_dist_shutdown(void);
int skdsp_dist_comm_count(int *nranks);           /* ncclCommCount of the live communicator; 0 = none (1-rank job) */
int skdsp_dist_barrier(void);                     /* 1-element RCCL all-reduce + stream sync */
int skdsp_dist_allreduce_max(double *value);      /* in-place max over ranks */
int skdsp_dist_allreduce_sum(double *value);
/* one grouped RCCL point-to-point step on the library stream: send `bytes` to rank dst and
 * receive `bytes` from rank src (either may be -1 = none).  The halo exchange is built on it. */
int skdsp_dist_sendrecv(const void *send_dev, int dst, void *recv_dev, int src, int64_t bytes);
/* every rank contributes `bytes` bytes; recv_dev holds world*bytes, rank-major (IIR end states:
 * sharding.ShardedIIR).  One rank / no communicator: a device copy. */
int skdsp_dist_allgather(const void *send_dev, void *recv_dev, int64_t bytes);
/* Halo exchange for a contiguous sample-block shard: send my LAST n_halo samples of
 * x_dev (n local samples) to rank+1, receive rank-1's into x_dev[-n_halo..-1]
 * (rank 0 zero-fills).  x_dev must have n_halo samples of headroom before it. */
int skdsp_dist_halo_exchange(void *x_dev, int64_t n, int64_t n_halo, int dtype);
/* sharded .filter: halo exchange of ntaps-1 samples, then the local filter. */
int skdsp_fir_filter_shard_dev(skdsp_handle h, void *x_dev, int64_t n_local, void *y_dev);

#ifdef __cplusplus
}
#endif
#endif /* SKDSP_H */
