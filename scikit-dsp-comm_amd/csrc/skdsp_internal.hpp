// Internal declarations shared by the HIP translation units of libskdsp_hip.so.
// gfx950 (MI355X / CDNA4) only -- no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <mutex>
#include "../../include/skdsp.h"
#include "careful.hpp"

namespace skdsp {

// ------------------------------------------------------------------ errors
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define SK_HIP(call)                                                        \
    do {                                                                    \
        hipError_t _e = (call);                                             \
        if (_e != hipSuccess) return skdsp::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define SK_CHECK(cond, code, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            skdsp::set_error(__VA_ARGS__);   \
            return (code);                   \
        }                                    \
    } while (0)

// Which engines the calling thread's last API calls launched (tests assert that a shape reaches the engine they mean to exercise):
// every *_launch appends its name once; skdsp_debug_path() reads and clears.  A few bytes of thread-local state, no device work.
void note_path(const char *engine);

// ------------------------------------------------------------------ options
// Run-time switches.  Read from the environment ONCE (first use: SKDSP_<NAME>), changed afterwards only through
// skdsp_set_option(): no getenv() on any launch path.  Most are developer A/B switches of measured alternatives.
struct Options {
    int device = -1;          // SKDSP_DEVICE: GPU to bind when skdsp_init() was not called explicitly
    int fir_algo = 0;         // SKDSP_FIR_ALGO = auto|direct|ols (0|1|2): override of every handle's choice
    int dn_no_ols = 0;        // .dn never through the overlap-save decimating store
    int fir_mm = 1;           // 0: no matrix-pipe FIR kernels at all (register sliding-window kernels instead)
    int fir_bx = 1;           // 0: no bf16x3 matrix-pipe kernel (FP32 matrix pipe instead)
    int fir_bx_t16 = 1;       // the float32 plain filter on the matrix pipe stores its 256-output tiles as runs (a 4 x 4 transpose between registers and lane groups); 0: four 64-byte runs per store (A/B switch)
    int ols_keep_overlap = 1; // overlap-save tiles: the blocks a tile shares with its neighbours are loaded with ordinary (L2-resident) loads, the rest nontemporal; 0: all nontemporal (A/B switch)
    int ols_reserve = 8;      // workgroup slots a persistent overlap-save launch leaves free
    int iir_planar = 0;       // complex IIR through two real planes (tests compare it with the interleaved kernels)
    int iir_dn_full = 0;      // .dn as full-rate scan + downsample kernel
    int iir_no_mfma = 0;      // recurrence K1 instead of the matrix-pipe K1
    int iir_two_pass = 0;     // 1: K1 + carries + K3 even where the single-pass scan applies; -1: single pass wherever it applies
    int iir_par = 1;          // 0: never the parallel-form scan (iir_par.hip); the cascade kernels everywhere
    int iir_par_v32 = 1;      // the parallel form's from-rest end states of float32 / complex64 signals, 7 - 8 biquads, on the float32 matrix instruction: 1 where the
                              // plan's probe admits the filter (iir_par.hip: par_v32_probe), 2 always (tests, A/B), 0 never
    int iir_up_jump = 1;      // the parallel-form .up of float32 / complex64 signals by L >= 8, a divisor of 96: lean kernels whose state jumps from input sample to input sample; 0 never (A/B switch)
    int iir_seq = 1;          // cascades of more than 8 sections whose float64 spread the scans would lift past the contract run the reference's recursion (iir_seq.hip): 1 probed, 2 always, 0 never
    int iir_up_lean = 1;      // multirate_IIR.up by 2 staged at the input rate with the stuffed zeros known at compile time (A/B switch; 0: the zero-stuffed image)
    int iir_dn_t96 = 1;       // the parallel-form .dn of float32 / complex64 signals on 96-sample chunks: 1 where measured to pay (see iir_par_launch), 2 wherever M divides 96, 3 as 1 but M = 2 keeps its gathering in ranges for every cascade, 0 never (A/B switch)
    int iir_dn_compact = 1;   // 0: the parallel-form .dn keeps the image-and-pick store for every M (A/B switch)
    int fir_up_ols_min = 64;  // multirate_FIR.up: phases of at least this many taps MAY go through the overlap-save walk (the cost model
                              // of fir_up_prefers_ols decides); 0: never; -k: always from k taps per phase on (A/B switch)
    int fir_up_rows_min = -1; // multirate_FIR.up through the overlap-save walk: from this L on the phases leave as rows and a second kernel weaves them
                              // (-1: the measured crossover per dtype, fir_up_rows in capi.hip; 0: never)
    int fir_up_pair = 1;      // 0: float32 .up through the overlap-save walk never pairs its phases (A/B switch)
    int fir_up4k = 1;         // 0: multirate_FIR.up never through the one-workgroup-per-input-tile interpolator (fir_up4k.hip); the older engines instead (A/B switch)
    int fir_up2k = 1;         // the 2048-point tile with all phases per thread (fir_up2k.hip): 1 from five passes on (complex64: L >= 5, float32: L >= 9), 2 always, 0 never (A/B switch)
    int fir_up_rep = 1;       // multirate_FIR.up, even L, through the replicated spectrum of the zero-stuffed tile (ols_rep_kernel): 1 where the cost model prefers it, 2 wherever it applies, 0 never (A/B switch)
    int fir_dn_fold = 1;      // multirate_FIR.dn through overlap-save, M = 2, 4, 8, 16: the folded spectrum's inverse transform (ols_fold_kernel); 0: the decimating store (A/B switch)
    int fir_dn4k = 1;         // multirate_FIR.dn through the frequency-domain decimator (fir_dn4k.hip): 1 where the cost model prefers it, 2 wherever it applies, 0 never (A/B switch)
    int fir_up4k_group = 4;   // phases (float32: pairs of phases) whose results a thread of that kernel holds before it stores: 4 (32 bytes per lane) or 2 (A/B switch)
    int fir_up4k_staged = 1;  // 0: four-pass groups of that kernel store each lane's own 32 bytes (A/B switch)
    int fir_updn_fused = 1;   // 0: L / M through the overlap-save walk writes all n L outputs to scratch and copies every M-th (A/B switch)
    int iir_up_fused = 1;     // 0: multirate_IIR.up / rate_change.up write the zero-stuffed signal first (A/B switch)
    int shard_two_launches = 0; // sharded FIR: tile 0 as its own launch behind the halo event (instead of the in-kernel flag wait)
    int shard_probe = 1;        // 1: a process's FIRST sharded step runs the two-launch form, its second the overlapped form on probation (short
                                // poll, tile 0 repeated behind the halo event, one sync) and only a passed probation makes the overlapped form
                                // the steady state (dist.hip); 0: overlapped from the first step; 2 (test hook): the probation step's flag is
                                // never published, i.e. the probation fails
    int shard_halo_state = 0;   // read-only mirror of that state machine: 0 no sharded step yet, 1 first step done (probation next), 2 overlapped form
                                // proven, 3 probation failed (two-launch form for good)
    int shard_self_halo = 0;    // test hook: a 1-rank communicator sends its tail to ITSELF (exercises the whole halo path on one GPU)
    int dist_force_comm = 0;  // build an RCCL communicator for a 1-rank job too (exercises the plumbing on one GPU)
    int host_chunk_log2 = 22; // host-pointer entry points: samples per pipelined chunk (pinned double buffers)
    int host_pipeline = 1;    // 0: single staged copy in / kernel / copy out
    int host_multi_slot = 1;  // 0: host-pointer FIR calls stay on the caller's slot even when several are bound
};
Options &opt();

// ------------------------------------------------------------------ context
// One Context per SLOT: a (device, stream, workspaces) binding.  Slot 0 is what every caller thread uses (skdsp_init);
// skdsp_init_devices binds further slots -- one per GPU of the node, or several on one GPU -- which the host-pointer
// entry points use from their own worker threads to spread a long host vector over all of them.  Each slot has its own
// lock: calls that run on different slots do not serialise each other.
constexpr int kMaxSlots = 16;
struct HostPipe;  // capi.hip: pinned-free chunk pipeline state (streams, events, double buffers)
struct Context {
    bool ready = false;
    int slot = 0;
    int device = -1;
    HostPipe *pipe = nullptr;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // grow-only device workspaces used by the host-pointer entry points
    hipStream_t comm_stream = nullptr;   // halo traffic of a sharded FIR, overlapped with the interior tiles
    hipEvent_t ev_in = nullptr, ev_halo = nullptr;
    unsigned *halo_flag = nullptr;       // device word the halo stream bumps when a shard's history has landed
    unsigned *async_err = nullptr;       // host-mapped [kAsyncErrWords]: kernels of this slot report here (async_err_check)
    unsigned halo_seq = 0;
    double last_timer_ms = -1.0;         // skdsp_last_kernel_ms
    int halo_state = 0;                  // the probation state machine of the sharded FIR step (Options::shard_halo_state mirrors it)
    void *ws[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t ws_bytes[4] = {0, 0, 0, 0};
    std::mutex mu;
};
// Failures a kernel can only report after the fact (a bounded device-side wait that gave up) land in host-mapped words of
// the slot; every entry point that synchronises the slot's stream tests them right after the sync, so the error is
// returned by the call that makes the affected results visible (and the word is cleared: later calls are not poisoned).
enum { kAsyncErrHalo = 0, kAsyncErrIirLookback = 1, kAsyncErrWords = 4 };
unsigned *async_err_dev(int which);   // device alias of the calling slot's word (allocated on first use); null on failure
int async_err_check(Context &c);      // call after a stream sync of slot c
Context &ctx();               // the calling thread's current slot (slot 0 unless a library worker selected another)
Context &ctx_of(int slot);
int slot_count();             // bound slots (0 before the first init)
int select_slot(int slot);    // thread-local: subsequent ctx() / launches of this thread use that slot (and its device)
int ensure_init();
int ws_reserve(int slot, size_t bytes, void **out);  // grow-only workspace slot 0..3

inline size_t dtype_size(int dt) { return dt == SKDSP_F32 ? 4 : dt == SKDSP_C64 ? 8 : dt == SKDSP_F64 ? 8 : 16; }
inline bool dtype_complex(int dt) { return dt == SKDSP_C64 || dt == SKDSP_C128; }
inline bool dtype_double(int dt) { return dt == SKDSP_F64 || dt == SKDSP_C128; }
inline bool dtype_valid(int dt) { return dt >= 0 && dt <= 3; }

// headroom (in samples) the host-pointer paths keep in front of the staged input
// so that kernels may be handed n_hist > 0; also keeps x 256-byte aligned.
constexpr int64_t kHeadroomBytes = 65536;

// ------------------------------------------------------------------ handles
enum HandleKind { H_FIR = 1, H_IIR = 2 };

struct HandleBase {
    int kind;
    int dtype;
    int slot = 0;           // slot whose device holds this handle's tables
    std::vector<HandleBase *> clones;  // same filter on the other slots, made on demand by the multi-slot host path (owned)
    bool wide_out = false;  // host-pointer entry points deliver float64/complex128 (skdsp_set_wide_output)
    std::mutex mu;
    virtual ~HandleBase()
    {
        for (HandleBase *c : clones) delete c;
    }
};

// ---- FIR -----------------------------------------------------------------
struct OlsPlan;    // fir_ols.hip
struct Ols64Plan;  // fir_ols64.hip
struct FirHandle : HandleBase {
    int ntaps = 0;
    bool taps_complex = false;
    int algo = SKDSP_FIR_AUTO;
    std::vector<double> taps_host;  // ntaps (real) or 2*ntaps (complex, interleaved)
    void *taps64_dev = nullptr;     // the same on the device, natural order (careful.hpp: the rare exact path of a poisoned tile); lazy
    // polyphase tap banks, keyed by L (lazy): bank[phase][t] = b[phase + L*t], T = ceil(P/L)
    struct Poly { int L; int T; void *dev; };
    std::vector<Poly> poly;
    // sliding-window tap tables, keyed by (L, M, R)
    struct SwTab { int L, M, R; void *taps; void *rho; };
    std::vector<SwTab> sw;
    // Toeplitz-product (matrix pipe) A-operand tables, keyed by (L, M)  -- fir_mm.hip
    struct MmTab { int L, M, Lp, q, DS, RS, U0, K4; void *At; };
    std::vector<MmTab> mm;
    // bf16x3 Toeplitz-product A-operand tables, keyed by (L, M)  -- fir_bx.hip
    struct BxTab { int L, M, Lp, q, DS, RS, RT, U0, KB, RSP, KSP; float tap_inv; void *At; };   // RT / KB: row tiles / 32-lag blocks of the table; RSP / KSP: waves they are dealt to
    std::vector<BxTab> bx;
    OlsPlan *ols = nullptr;
    struct OlsUp { int L; OlsPlan *plan; };   // overlap-save plans of multirate_FIR.up, keyed by L (fir_ols_up_launch)
    std::vector<OlsUp> ols_up;
    void *up4k = nullptr;   // plans of the frequency-domain interpolator, keyed by L (fir_up4k.hip)
    void *up2k = nullptr;   // ... of its many-phase form (fir_up2k.hip)
    void *dn4k = nullptr;   // plans of the frequency-domain decimator, keyed by M (fir_dn4k.hip)
    Ols64Plan *ols64 = nullptr;
    struct Ols64Up { int L; Ols64Plan *plan; };
    std::vector<Ols64Up> ols64_up;
    // Filters longer than one kernel launch takes (fir_part_len) run as partial FIRs over consecutive tap segments,
    // each applied to the correspondingly delayed input and summed (capi.hip): parts[s] holds taps [s seg, (s+1) seg).
    std::vector<FirHandle *> parts;
    int part_seg = 0;
    // Calls FROM REST over fewer samples than taps only ever reach the first n taps: they run on a copy of the filter cut to the next power of
    // two >= n (capi.hip, fir_head) -- exact, and the float32 rounding then scales with the taps that matter, not with the whole filter
    std::vector<FirHandle *> heads;
    ~FirHandle();
};

// the taps as the careful path reads them (uploaded on first use, under the handle's lock like every other table)
int fir_careful(FirHandle *h, CarefulFir *out);

// direct-form / polyphase launcher (fir_direct.hip).
//   y[m] = L * sum_t b[phi + L t] * x[i - t],  j = m*M, phi = j mod L, i = j div L,  m in [0, n_out)
int fir_direct_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, int M, int64_t n_out,
                      void *y_dev, hipStream_t s);
// Toeplitz product on the FP32 matrix pipe (fir_mm.hip): float32 / complex64 signals, real taps
bool fir_mm_supported(const FirHandle *h, int L, int M, int64_t n_out);
int fir_mm_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, int M, int64_t n_out, void *y_dev,
                  hipStream_t s);
// Toeplitz product on the BF16 matrix pipe in float32 precision (3-way bf16 split, fir_bx.hip): same coverage, tried first
bool fir_bx_supported(const FirHandle *h, int L, int M, int64_t n_out);
int fir_bx_blocks(const FirHandle *h, int L, int M, int *row_tiles = nullptr);  // 32-lag blocks per output tile, 0 = not covered
int fir_bx_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, int M, int64_t n_out, void *y_dev,
                  hipStream_t s);
// FFT overlap-save (fir_ols.hip): c64 (and packed f32) .filter
bool fir_ols_supported(const FirHandle *h);
int fir_ols_tile_outputs(FirHandle *h, int *V);  // outputs per overlap-save tile (builds the plan if needed)
int fir_algo_for(const FirHandle *h, int64_t n);  // SKDSP_FIR_OLS / SKDSP_FIR_DIRECT as skdsp_fir_filter_dev would pick
int fir_ols_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, void *y_dev, hipStream_t s,
                   int dec = 1,           // dec > 1: decimating store, y holds n / dec samples
                   int reserve_wgs = -1,  // persistent grid smaller by this many workgroups (multiple of 8; -1: default 8)
                   // sharded launches: tile 0 is walked last and waits until *halo_flag >= halo_seq (the history in front
                   // of x is being received on another stream); halo_err (host-mapped) is set if that wait gives up
                   const unsigned *halo_flag = nullptr, unsigned halo_seq = 0, unsigned *halo_err = nullptr,
                   int halo_spins = 0);   // polls of ~1 us before that wait gives up (0: the steady-state bound, seconds)
int fir_ols_publish_halo(unsigned *flag, unsigned seq, hipStream_t s);  // one-thread kernel: *flag = seq (agent-scope release)
void fir_ols_free(OlsPlan *p);
// multirate_FIR.up as an overlap-save walk over (tile, phase) pairs: complex64, float32 with real taps; 2..4097 taps per phase
bool fir_ols_up_supported(const FirHandle *h, int L);
bool fir_ols_up_pairs(const FirHandle *h, int L, int dec, const void *y_dev);   // float32, even L: phases in pairs through the complex tile (8-byte outputs)
int fir_ols_up_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, void *y_dev, hipStream_t s, int dec = 1,
                      int64_t rows_pitch = 0, int paired = 0);  // dec = M: L / M, floor(n L / M) outputs; rows_pitch > 0: y[phase * rows_pitch + i] instead of
                                                                 // y[i L + phase]; paired: see fir_ols_up_pairs (rows then hold 8-byte pairs, L / 2 of them)
// multirate_FIR.up, even L, tiles of the OUTPUT: the zero-stuffed tile's forward transform from its non-zero columns (fir_ols.hip: ols_rep_kernel)
bool fir_ols_rep_supported(const FirHandle *h, int L);
int fir_ols_rep_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, void *y_dev, hipStream_t s);
// multirate_FIR.up, one workgroup per input tile, all L phases from one forward transform (fir_up4k.hip): complex64, float32 with real
// taps; at most 2049 taps per phase
bool fir_up4k_supported(const FirHandle *h, int L);
int fir_up4k_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, void *y_dev, hipStream_t s);
void fir_up4k_free(void *plans);
// the same with a 2048-point tile and ALL phases of a sample in one thread (fir_up2k.hip): at most 1025 taps per phase
bool fir_up2k_supported(const FirHandle *h, int L);
int fir_up2k_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, void *y_dev, hipStream_t s);
void fir_up2k_free(void *plans);
// multirate_FIR.dn, one workgroup per OUTPUT tile: M forward transforms accumulated in the frequency domain, one inverse (fir_dn4k.hip)
bool fir_dn4k_supported(const FirHandle *h, int M);
int fir_dn4k_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int M, void *y_dev, hipStream_t s);
void fir_dn4k_free(void *plans);
// FFT overlap-save in float64 (fir_ols64.hip): complex128, and float64 with real taps; 2..2049 taps
bool fir_ols64_supported(const FirHandle *h);
int fir_ols64_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, void *y_dev, hipStream_t s, int dec = 1);
void fir_ols64_free(Ols64Plan *p);
bool fir_ols64_up_supported(const FirHandle *h, int L);
bool fir_ols64_up_pairs(const FirHandle *h, int L, int dec, const void *y_dev);
int fir_ols64_up_launch(FirHandle *h, const void *x_dev, int64_t n, int64_t n_hist, int L, void *y_dev, hipStream_t s, int dec = 1, int64_t rows_pitch = 0, int paired = 0);

// ---- IIR -----------------------------------------------------------------
struct IirPlan;  // iir_scan.hip
struct IirHandle : HandleBase {
    int nsec = 0;   // number of cascaded sections
    int order = 0;  // order of each section (2 for SOS, K-1 for a transfer function)
    std::vector<double> coef;  // per section: b[0..order], a[1..order]  (a0-normalised)
    // Biquad cascades whose sections all have b2 == b0 (zeros on the unit circle: every Butterworth / Chebyshev /
    // elliptic design of scipy.signal, all band types) are re-factored at creation -- the same transfer function with
    // all gain in section 0 and b0 = b2 = 1 in the others -- so that K3 needs 3 coefficients and 4 flops per such
    // section (iir_scan.hip).  state_scale[d] converts a DF2T state of the caller's factorisation into the internal
    // one (z_int = scale * z); empty = identity.
    bool unit_tail = false;
    std::vector<double> state_scale;
    IirPlan *plan = nullptr;
    struct ParPlan *par = nullptr;   // partial-fraction form of the same transfer function (iir_par.hip), made on first use
    // Cascades of more than 8 biquads run as consecutive groups of at most 8 (the kernels' register budgets; 8 is what the parallel form
    // takes): group g filters the output of group g - 1 in place, each with its own plans; the caller's per-section states are sliced.
    // scipy.signal.sosfilt takes any number of sections, and so does this.
    std::vector<IirHandle *> groups;
    // float32 handles whose sections cannot be grouped in float32 (the rounding of the signal between two groups, amplified by the rest
    // of the cascade, would break the float32 contract: capi.hip) and are too many for one launch sequence: the same cascade as a float64
    // handle; the signal is widened, filtered and narrowed on the device
    IirHandle *twin64 = nullptr;
    void *twin_in = nullptr, *twin_out = nullptr;
    size_t twin_in_bytes = 0, twin_out_bytes = 0;
    // Cascades whose float64 result is itself uncertain beyond what the scans may add to it (capi.hip: the probe at creation) run the reference's
    // own recursion (iir_seq.hip): the caller's sections [nsec][5], the relative float64 spread the probe measured
    bool seq = false;
    double seq_spread = 0.0;
    std::vector<double> seq_coef;
    void *seq_coef_dev = nullptr;
    int group_first = 0;             // (a group: its first section in the parent's numbering)
    void *group_tmp = nullptr;       // full-rate intermediate of a decimating call
    size_t group_tmp_bytes = 0;
    ~IirHandle();
};
// x/y: real planar arrays in the handle's precision; complex callers pass nbatch=2 planes
// (re, im) batch_stride elements apart.  In-place (y == x) is allowed.
int iir_launch_planar(IirHandle *h, const void *x_dev, int64_t n, int nbatch, int64_t batch_stride, void *y_dev, hipStream_t s,
                      const double *zi_host = nullptr, double *zf_host = nullptr,  // [nbatch][D] states (streaming)
                      int interleaved = 0,   // 1: x / y interleaved complex, nbatch = 2; returns 1 if not applicable
                      int dec = 1);          // > 1 (real signals): y receives only every dec-th output (n / dec samples)
void iir_free(IirPlan *p);
// the reference's recursion, one wave per row (iir_seq.hip): handles with h->seq set
int iir_seq_launch(IirHandle *h, const void *x_dev, int64_t n, int nrow, int64_t x_stride, int64_t y_stride, void *y_dev, hipStream_t s,
                   const double *zi_host = nullptr, double *zf_host = nullptr, int dec = 1);
// Parallel-form single-pass scan (iir_par.hip): real signals, <= 8 biquads with simple poles, zero initial state, no state
// output; nrow rows x_stride / y_stride elements apart in one launch.  Returns 1 when it does not apply (nothing launched).
int iir_par_launch(IirHandle *h, const void *x_dev, int64_t n, int nrow, int64_t x_stride, int64_t y_stride, void *y_dev, hipStream_t s,
                   int dec = 1, int interleaved = 0,    // interleaved = 1: x / y interleaved complex, n complex samples, one row
                   int up = 1);                         // up > 1: x holds n / up samples, the launch filters up * upsample(x, up) (n outputs)
int iir_par_expand_host(const double *coef, int nsec, double *out, int *accepted);   // host-only (tests): [c0, (a1,a2,r0,r1) x nsec, kappa, ir_err]
void iir_par_free(ParPlan *p);
bool iir_shape_supported(int nsec, int order);

// ---- resamplers (resample.hip) ---------------------------------------------
int upsample_launch(const void *x_dev, int64_t n, int L, int dtype, double scale, void *y_dev, hipStream_t s);
int upsample_planes_launch(const void *x_dev, int64_t n, int L, int dtype_complex_in, double scale, void *re_dev, void *im_dev,
                           hipStream_t s);  // complex zero-stuffing straight into two real planes
int downsample_launch(const void *x_dev, int64_t n, int M, int p, int dtype, void *y_dev, hipStream_t s);
int interleave_launch(const void *src_dev, int64_t n, int L, int64_t pitch, int dtype, void *y_dev, hipStream_t s);   // y[i L + p] = src[p pitch + i]
int deinterleave_launch(const void *x_dev, int64_t n, int dtype_complex_in, void *re_dev, void *im_dev, hipStream_t s);
int interleave_launch(const void *re_dev, const void *im_dev, int64_t n, int dtype_complex_out, void *y_dev, hipStream_t s);
int widen_launch(const void *src_dev, int64_t nscalars, void *dst_dev, hipStream_t s);  // float32 -> float64
int convert_launch(const void *src_dev, void *dst_dev, int64_t nscalars, bool to_double, hipStream_t s);   // element-wise float32 <-> float64, any alignment
int accumulate_launch(void *y_dev, const void *t_dev, int64_t nscalars, bool dbl, hipStream_t s);  // y += t
int fill_noise_launch(void *x_dev, int64_t n, int dtype, uint64_t seed, int64_t first, hipStream_t s);

}  // namespace skdsp
