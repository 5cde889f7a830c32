// iir_par.hip -- single-pass exact IIR scan in PARALLEL FORM for gfx950 (MI355X): one launch, the signal read once and
// written once, every wavefront an independent segment.
// Serves scipy.signal.sosfilt(sos, x) (multirate_helper.py:173, :181, :190) / lfilter(b, a, x) (:74, :81) for decaying
// cascades of up to 8 biquads with simple poles when no state crosses the call (zi / zf callers keep the cascade kernels
// of iir_fused.hip / iir_scan.hip, whose state coordinates are scipy's).
//
// Why a second formulation.  The cascade runs 5 (4 in unit-tail form) DEPENDENT FP64 instructions per biquad and sample --
// 40 per sample for 8 biquads, one chain -- and its one-chunk transition is a dense block-triangular 16 x 16 matrix, so
// every level of the chunk scan is a 16 x 16 matrix-vector product per chunk.  The same transfer function expanded in
// partial fractions,
//     H(z) = c0 + sum_k (r0_k + r1_k z^-1) / (1 + a1_k z^-1 + a2_k z^-2),
// is 8 INDEPENDENT two-state recurrences (2 FMA each) plus a 17-term output sum that depends on the previous states only:
// 33 FP64 instructions per sample with no chain longer than two, and a block-DIAGONAL transition -- a scan level is 8
// 2 x 2 products (32 FMA) instead of 144-256.  The expansion is done on the host in long double by arithmetic modulo each
// section's denominator (no root finding, so a section may hold a complex pair, two real poles or a double pole); it is
// accepted only when the impulse response of the expansion, with its coefficients rounded to double, reproduces the cascade's to
// 1e-12 and its branches do not cancel (sum of the branch l1 norms <= 1e3 x the l1 norm of h): every design scipy makes for
// rate_change / IIR_bpf / IIR_lpf passes with a factor of 2-20, i.e. a roundoff of ~1e-14 of the output.
//
// Structure (real signals).  A WAVE owns one segment of 64 chunks x T samples (T = 128 float32 / 64 float64: 8192 / 4096
// samples) and keeps it on chip -- each lane its own chunk in registers:
//   A  the segment streams in through a wave-private [64 rows x 32 samples] LDS image (full-line 16-byte loads, next piece in
//      flight); each lane copies its row into registers and the same image feeds the FP64 matrix pipe with the B operands
//      of V = G x, the from-rest end states (w[T-1], w[T-2]) of all 8 sections of the wave's 64 chunks
//      (v_mfma_f64_16x16x4_f64, 16 state rows = one tile height exactly)
//   S  from-rest Hillis-Steele scan of the wave's 64 chunk states: per level 8 ds_read_b128 + 32 FMA per lane, matrices
//      through the scalar cache; only levels whose power of the chunk transition is not negligible
//   L  decoupled look-back: the wave publishes its from-rest end state (32 eight-byte {half, epoch} granules, relaxed
//      agent-scope atomics: the data is its own flag) and reads those of its K predecessors.  K = the number of segments the
//      filter remembers (config 4: 1); the state at the start of the segment is c = sum_j Psi^j P_(s-1-j), exact to the
//      negligibility threshold, from FROM-REST values only -- no wave waits for a chain.  Segments are handed out by a
//      ticket counter, so a predecessor is a wave that already runs; the poll is bounded and reports through a host-mapped
//      word instead of hanging
//   C  z_j = p_(j-1) + Phi^j c by binary powers
//   B  the recurrence over the register-resident samples; outputs leave through the LDS image as full lines
// Nothing in A..B meets another wave: no workgroup barrier after the table load, so the 8 waves of a CU drift through
// their memory / matrix / LDS / VALU phases independently.  N-D inputs are rows of the same launch (row = ticket / nseg).
#include "iir_common.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

namespace skdsp {

constexpr int kParMaxK = 8;   // look-back depth served (segments the filter may remember): 8 for float64 signals, 4 for float32 ones
// (half the samples per segment and a 1e-30 instead of a 1e-18 threshold make a float64 segment "shorter" for the same filter; the look-back
// words of K segments live behind the scan exchange inside the wave's stage image, which is twice as large for float64)
constexpr int par_max_k(bool dbl) { return dbl ? 8 : 4; }
constexpr int kParTickets = 16;   // segment dispensers per launch
constexpr int SK_PAR_T32 = 128;   // samples per lane (float32 signals); float64 signals: half
constexpr int SK_PAR_PRIO = 3;      // wave priority in front of the recurrence (float32 signals, 6 - 8 biquads: DESIGN.md)
constexpr int SK_PAR_PRIO_ST = 1;   // ... and on a piece's way out through the image
constexpr int SK_PAR_OCC = 2;     // waves per SIMD the register budget is set for
constexpr int kParStageM2 = 13312;  // bytes of a wave's stage image in the DECM = 3 kernels (3072 + 3072 / 32 + 2 slots of 4 bytes, rounded up)
// (measured and not kept, round 6: three waves per SIMD for the lean decimating kernel with float32 from-rest states -- 168 VGPRs, 148 bytes of scratch per
// lane: 0.135 against 0.133 ms at two; the recurrence alone holds 96 sample + 32 state + 32 tap registers)
constexpr int SK_PAR_OCC_UPL = 4;   // ... of the lean .up kernels (UPJ, up to 4 biquads: no input image, no table in LDS)

template <int NSEC> struct ParCoef {
    double na1[NSEC], na2[NSEC];   // -a1, -a2
    double al[NSEC], be[NSEC];     // output taps on (w[n-1], w[n-2])
    double gamma;                  // direct term
};

struct ParArgs {
    const void *x;
    void *y;
    int64_t n;                   // samples per row
    int64_t x_stride, y_stride;  // elements between rows
    int nseg;                    // segments per row
    int total;                   // rows x nseg
    unsigned long long *lb;      // [total][32] look-back granules
    unsigned long long *ticket;  // [kParTickets] segment dispensers (monotonic; ticket_base = their common value before this launch)
    unsigned long long ticket_base;
    unsigned epoch;
    int n_lv;                    // scan levels inside a wave that matter (0..6)
    int K;                       // look-back depth (1..kParMaxK)
    unsigned *err;               // host-mapped: a look-back poll gave up
    int aligned;                 // rows start 16-byte aligned (vector loads / stores)
    int dec;                     // > 1: only y[k * dec] is stored (at y[k])
    int dec_dq, dec_dr;          // (rows between a lane's staged segments x T) div / mod dec
    int64_t n_keep;              // (n / dec) * dec
    // dec_compact: the kept outputs of a whole segment are gathered in the wave's (idle) stage image and leave as one contiguous run of
    // y (dec >= the samples of a 16-byte unit, and n_seg / dec of them fit the image); else the image-and-pick path below
    int dec_compact;
    // dec_rounds > 1 (float32 / complex64 with dec = 2, 3: a segment keeps more than the image holds, and a 16-byte unit up to two samples):
    // the gathering happens behind the recurrence, dec_rounds ranges of chunks one after the other, each range a run of its own
    int dec_rounds;
    unsigned dec_magic;          // ceil(2^32 / dec)
    // up > 1 (one row): x holds n_in = n / up samples and the kernel filters up * upsample(x, up) -- the zero-stuffed
    // signal exists only in the wave's staging image (multirate_IIR.up / rate_change.up: multirate_helper.py:69-75, 177-184)
    int up;
    unsigned up_magic;           // ceil(2^32 / up): (v * up_magic) >> 32 = v / up for the v < 2^15 met here
    int64_t n_in;
};

// v += P * left, P = the level's 2 x 2 block of every section (row-major), through the scalar cache
template <int NSEC>
__device__ __forceinline__ void blocks_acc(const double *__restrict__ P, const double (&in)[2 * NSEC], double (&out)[2 * NSEC])
{
#pragma unroll
    for (int k = 0; k < NSEC; ++k) {
        out[2 * k] = fma(P[4 * k + 1], in[2 * k + 1], fma(P[4 * k], in[2 * k], out[2 * k]));
        out[2 * k + 1] = fma(P[4 * k + 3], in[2 * k + 1], fma(P[4 * k + 2], in[2 * k], out[2 * k + 1]));
    }
}

typedef double v2d_t __attribute__((ext_vector_type(2)));

// DECM: 0 every output is stored; 1 .dn (the kept outputs gathered per piece, or picked out of the image); 2 .dn with dec = 2, 3 on 4-sample
// units (gathered behind the recurrence in ranges of chunks) -- an instantiation of its own: inside the others its 64 predicated LDS
// writes cost the 8-biquad kernels 450 spilled SGPRs and 2 % on every other M
// TT: samples per chunk (0: SK_PAR_T32 / its half for float64).  The decimating kernels of float32 / complex64 signals run on chunks of 96 where M divides 96
// (M = 2, 3, 4, 6, 8, 12, 16, 24, ...): every chunk of every segment then starts on a kept sample, all lanes of a wave walk the SAME phase, and the 2 NSEC + 1
// term output sum is formed for one sample in M -- with 128, M = 3 put the lanes on three phases and every sum was formed (M = 12: three in twelve).
// UPJ (.up by 8 or more, with TT = 96 and L a divisor of 96): between two input samples the filter runs on stuffed zeros, so the recurrence does not step
// through them -- every output is a 2 NSEC term product of the state right behind the last input sample with a row of c A^j (a table, wave-uniform
// because all chunks start on an input sample), and the state jumps by A^L per INPUT sample: 2 NSEC + 5 NSEC / L multiply-adds per output instead of
// 4 NSEC + 1 (order-8 Butterworth, L = 12: 9.7 instead of 17).  float64 / complex128 signals too (they never hold the chunk in registers).
// UPS (.up by 2, 3, 4 -- the factors of sigsys.interp24's stages): the stuffed zeros are compile-time facts; staged at the input rate (UP2 in the kernel).
// The rate forms are LEAN: the bookkeeping of the general kernel (zero-stuffing unit by unit, per-lane phase tests, the pick of kept samples) was two
// thirds of their instructions -- see UPL / DNL / UP2 at the top of the kernel and LABNOTES R5.8.
// V32 (float32 / complex64 signals, 7 - 8 biquads, where the plan's probe accepts it: par_v32_probe): V = G x on v_mfma_f32_16x16x4_f32 -- half the issue cycles
// of the FP64 form (32 against 64; both share the vector datapath: tools/ubench_mixed_pipes.hip), G rounded to float32 once, the samples exact, a float32
// fmaf chain over the chunk, oldest sample first.  The from-rest end states then carry ~1e-7 of error, which the output sum amplifies by the cancellation
// between the branches -- so the form is EARNED per filter: the host runs the same chain bit for bit on probe inputs and admits the filter below 5e-7.
template <int NSEC, typename IO, int DECM, bool CPLX, int TT = 0, bool UPJ = false, int UPS = 0, bool V32 = false>
__global__ __launch_bounds__(kIirThreads, (UPJ && NSEC <= 4 && sizeof(IO) == 4) ? SK_PAR_OCC_UPL : SK_PAR_OCC) void iir_par_kernel(ParArgs a, ParCoef<NSEC> cf, const double *__restrict__ gtab,
                                                                 const double *__restrict__ lvl, const double *__restrict__ psi,
                                                                 unsigned long long,   // (keeps upj out of the register tuple the three pointers above arrive in: that tuple was spilled as a whole, upj with it, and restored -- eight registers -- in front of every pair of row loads)
                                                                 const double *__restrict__ upj = nullptr)   // UPJ: [up][2 nsec] rows c A^j (up to 4 biquads: row 0 once more), then [nsec][4] the blocks of A^up
{
    constexpr bool DEC = DECM != 0;
    constexpr int D = 2 * NSEC;
    // UPL, the lean form UPJ kernels take: a chunk of 96 outputs holds 96 / up <= 12 input samples and starts on one, so the zero-stuffed chunk
    // never exists, not even in the wave's image: a lane loads ITS inputs (adjacent lanes adjacent runs: whole lines per wave), forms its from-rest end
    // state from the 96 / up columns of G those meet (wave-uniform columns: scalar loads, no table in LDS) and hands the recurrence the next input as a
    // register.  The zero-stuffing through the image (two magic divisions, a dozen selects and an LDS round trip per 16-byte unit of OUTPUT-rate
    // samples) was two thirds of the kernel's 3000 vector instructions per segment (profiles/r05/pmc_rcup12.json: SQ_INSTS_VALU) -- the FP64 work is 1000.
    constexpr bool UP2 = UPS != 0;   // (named after the first of its factors)
    constexpr int UL = UPS ? UPS : 1;
    constexpr bool UPL = UPJ;   // (every UPJ kernel is lean; what differs by the number of biquads is how the rows of c A^j are requested)
    // UP2 (.up by 2 -- the smallest factor: one sample in two carries input and the state jump would cost more than it saves): every chunk starts on an input sample and every second sample is a stuffed zero -- known when
    // the kernel is compiled.  The segment is staged at the INPUT rate (the plain filter's loads over chunks of T / 2 samples: whole lines, no division, no
    // select), V = G x runs over the even columns of G, and the recurrence reads input k / 2 at even k and a literal zero at odd k.  Building the
    // zero-stuffed image unit by unit (two magic divisions, two loads, a dozen selects per 16 bytes of OUTPUT) was 2600 of the 7200 vector instructions per
    // segment of multirate_IIR(8 biquads).up(x, 2) -- the arithmetic is 4500 (SQ_INSTS_VALU, LABNOTES R5.8).
    // The same for 3 (chunks of 96) and 4 (float32 / complex64: a float64 chunk of 64 holds 16 inputs, half a staging piece).
    static_assert(!UP2 || (DECM == 0 && !UPJ && (UPS == 3 ? TT == 96 : TT == 0) && (UPS == 2 || UPS == 3 || UPS == 4)), "UPS: the plain store; by 3 on chunks of 96");
    // DNL, the lean form of the compact decimating store (TT = 96, M a divisor of 96 from 3 on: a segment's kept outputs fit the image): every chunk of every segment starts on a kept sample,
    // so WHICH samples are kept is wave-uniform -- a scalar counter and a scalar branch instead of the per-lane phase arithmetic (a compare and an exec
    // mask per sample, the unit bookkeeping per 16 bytes, the pick out of the unit at the gathering: 2000 of the 3700 vector instructions per segment of
    // rate_change(12).dn, profiles/r05/pmc_rcdn12.json).  A kept output goes straight from the sum into its slot of the wave's idle image.
    constexpr bool DNL = (DECM == 1 || DECM == 3) && TT == 96;   // (3: M = 2, whose 3072 kept outputs per segment need a larger image -- see kWaveStage)
    constexpr bool UNI = DEC && TT == 96;   // (M = 2 on 96-sample chunks keeps its gathering in ranges but tests for kept samples the same way)
    constexpr bool G4 = D <= 12;   // V = G x by 4 x 4 x 4 products over the row groups in use (see phase A)
    static_assert(!V32 || (sizeof(IO) == 4 && !G4 && !UPJ), "V32: float32 signals, more than 6 biquads, a kernel that runs V = G x on the matrix instruction");
    constexpr int T = TT ? TT : SK_PAR_T32 * 4 / (int)sizeof(IO);
    constexpr int NP = T / kPiece;
    constexpr int TI = T / UL;   // samples per chunk at the rate the signal is READ at
    constexpr int NPI = TI / kPiece;
    static_assert(TI % kPiece == 0, "UP2: whole pieces at the input rate");
    static_assert(T % kPiece == 0 && ((T / 4) * 64) % kIirThreads == 0, "chunk length: whole pieces, a table the workgroup loads evenly");
    using St = Stage<IO>;
    constexpr int kRowBytes = St::pitch * (int)sizeof(IO);
    // 9216 (float) / 17408 (double) bytes: also holds the 8 KiB scan exchange.  DECM = 3 (the lean compact store at M = 2, float32 / complex64, more than 4
    // biquads -- kernels that run two workgroups per CU anyway): room for the 3072 kept outputs of a segment and their padding
    constexpr int kWaveStage = DECM == 3 ? kParStageM2 : 64 * kRowBytes;
    constexpr int KMAX = par_max_k(sizeof(IO) == 8);
    static_assert(kWaveStage >= 64 * 16 * 8 + KMAX * 64 * 4, "scan exchange + look-back words must fit the wave's stage image");
    __shared__ __attribute__((aligned(16))) char lds_raw[4 * kWaveStage];
    __shared__ double gl[UPL ? 64 : (T / 4) * 64];
    __shared__ int base_sh;
    float *glf = reinterpret_cast<float *>(gl);   // V32: the table as float32 (the first half of the same array)
    // CPLX: an interleaved complex signal.  Lane L owns component L & 1 (re / im) of complex chunk L >> 1: T complex samples
    // per chunk, 32 chunks per wave segment.  The two components are independent real signals through the same real
    // filter, so everything between the staging image and the recurrence is the real kernel with "the chunk to my left"
    // two lanes away; only the staging (de-interleave on the way in, re-interleave on the way out) and the look-back
    // (two end states per segment) differ.
    constexpr int LS = CPLX ? 2 : 1;            // lanes per chunk
    constexpr int CH = 64 / LS;                 // chunks per wave segment
    constexpr int GR = 32 * LS;                 // look-back granules per segment

    // (the wave index through readfirstlane: segment, row and every base address are then wave-uniform SGPR values)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Everything in front of the recurrence runs at raised priority -- where the recurrence is what the launch waits for.  The
    // recurrence of the wave that shares this SIMD is a dense v_fma_f64 stream and, being the older wave, wins every issue
    // arbitration: a younger wave's G x products (the same FP64 datapath), its scan and its correction then crawl
    // (tools/par_trace.py: 29 k clocks for 128 MFMAs that take 8 k) and are still unfinished when the older wave's recurrence ends --
    // the pipe idles until they are.  With priority the preparation is over early and the next recurrence starts the moment the
    // pipe is free.  Config 4: 0.161 -> 0.150 ms float32, 0.322 -> 0.303 ms complex64 (same box, alternating); float64 signals and
    // cascades of fewer than 6 biquads are bound by their memory walk and LOSE 2-6 % with it (the stores of the older wave then
    // wait behind the younger wave's loads), so the switch follows the FP64 work per byte.
    constexpr bool PRIO = SK_PAR_PRIO != 0 && sizeof(IO) == 4 && NSEC >= 6;
    if (PRIO) __builtin_amdgcn_s_setprio(SK_PAR_PRIO);
    // kParTickets dispensers, one per workgroup residue (a single word serialises the 2048 draws of a 2^26-sample launch in
    // the L2 atomic unit: 22 us of an otherwise empty launch): workgroup b draws q from dispenser b mod kParTickets and serves
    // wave segments 4 (kParTickets q + b mod kParTickets) ...; every dispenser hands out exactly the numbers of its residue
    // class, so the launch covers every segment once, and the smallest segment not yet drawn is drawn by the next workgroup
    // of its class that starts -- which needs only that running workgroups finish, and those wait for smaller segments only
    // (the table is requested first and the ticket behind it, so that the two round trips overlap: thread 0 used to wait for its ticket
    // before it asked for its share of the table, and the barrier for thread 0)
    constexpr int kTabPer = (T / 4) * 64 / kIirThreads;
    double tab[kTabPer];
    if constexpr (!UPL) {
#pragma unroll
        for (int i = 0; i < kTabPer; ++i) tab[i] = gtab[tid + i * kIirThreads];
    }
    // UPL: the blocks of A^up in vector registers (used once in `up` samples: not worth 32 scalar registers).  Vector loads of one address -- a lane offset
    // the compiler cannot see through: as scalar loads they were 4 NSEC round trips one after the other, each waited for before its value could move over
    double AuV[UPL ? 4 * NSEC : 1];
    if constexpr (UPL) {
        int vz = 0;
        asm volatile("" : "+v"(vz));
        const double *Ap = upj + (size_t)(a.up + (NSEC <= 4 ? 1 : 0)) * D + vz;   // (up to 4 biquads the table repeats row 0 behind row up - 1)
#pragma unroll
        for (int i = 0; i < 4 * NSEC; ++i) AuV[i] = Ap[i];
    }
    unsigned long long drawn = 0;
    if (tid == 0) drawn = atomicAdd(a.ticket + blockIdx.x % kParTickets, 1ull);
    if constexpr (!UPL) {
#pragma unroll
        for (int i = 0; i < kTabPer; ++i) {
            if constexpr (V32) glf[tid + i * kIirThreads] = (float)tab[i];
            else gl[tid + i * kIirThreads] = tab[i];
        }
    }
    if (tid == 0) base_sh = 4 * (int)((unsigned)(drawn - a.ticket_base) * kParTickets + blockIdx.x % kParTickets);
    __syncthreads();   // the only workgroup barrier
    // (UPL, measured and not kept: PERSISTENT waves that draw their own tickets and ask for the next one before they start on a segment -- 0.072 -> 0.135 ms:
    // a ticket held early is a segment whose from-rest state appears a segment's time late, and its successor, drawn by another wave a moment later,
    // waits for it in the look-back.  The draw itself is 8 % of rate_change(12).up (ticket replaced by blockIdx: 0.0755 -> 0.0695 ms), which needs the
    // dispatch order the hardware happens to follow and promises nowhere.)
    // (Round 6, measured once more for the plain filter and in the form that draws ON TIME: every wave persistent, drawing segment 16 q + class from its class's
    // dispenser when it is free for it.  Config 4 0.143 -> 0.200 ms, .dn(x, 3) 0.135 -> 0.190: the classes drift apart, and a wave whose predecessor's class is a
    // round behind polls for it instead of working -- the four waves of a workgroup that start four CONSECUTIVE segments together are what keeps the look-back short.)
    const int tk = __builtin_amdgcn_readfirstlane(base_sh) + wave;
    if (tk >= a.total) return;
    const int row = a.nseg == a.total ? 0 : __builtin_amdgcn_readfirstlane(tk / a.nseg), seg = tk - row * a.nseg;
    const IO *x = reinterpret_cast<const IO *>(a.x) + (size_t)row * a.x_stride;
    IO *y = reinterpret_cast<IO *>(a.y) + (size_t)row * a.y_stride;
    IO *stage = reinterpret_cast<IO *>(lds_raw + wave * kWaveStage);
    double *E = reinterpret_cast<double *>(lds_raw + wave * kWaveStage);   // scan exchange [section][lane][2] (aliases the image)
    const int64_t row0 = (int64_t)seg * CH;   // first chunk of the segment
    const bool interior = a.aligned && (row0 + CH) * T <= a.n;   // (a.n: samples per row -- complex samples for CPLX)

    typedef float pre_t __attribute__((ext_vector_type(4)));
    pre_t pre[NP][St::per_thread];
    // (uniform segment base + one 32-bit lane offset + constants: hipcc then addresses every access of the segment as
    // SGPR base + VGPR offset + immediate instead of keeping a 64-bit address pair per access alive)
    const IO *xseg = x + row0 * TI * LS;         // (IO scalars: an interleaved complex sample is two)
    IO *yseg = y + row0 * T * LS;
    // a 16-byte unit of a chunk's piece: St::segs units per real chunk piece, 2 x St::segs per complex one (re/im interleaved)
    constexpr int USEG = St::segs * LS;
    const unsigned loff = (unsigned)((lane / USEG) * T * LS + (lane % USEG) * St::elems);
    constexpr unsigned kRowStep = (64 / USEG) * T * LS;   // IO scalars between a lane's consecutive staged units
    const unsigned loff_in = UP2 ? (unsigned)((lane / USEG) * TI * LS + (lane % USEG) * St::elems) : loff;   // (the same on the way in, at the rate read at)
    constexpr unsigned kRowStepIn = (64 / USEG) * TI * LS;
    auto load_piece = [&](int p) {  // interior segments only
#pragma unroll
        for (int i = 0; i < St::per_thread; ++i)
            pre[p][i] = __builtin_nontemporal_load(reinterpret_cast<const pre_t *>(xseg + (loff_in + i * kRowStepIn + p * kPiece * LS)));
    };
    // image position of staged unit (i, lane): real: 16 bytes of row idx / segs; complex: the unit holds elems / 2 complex
    // samples of chunk idx / USEG -- their re parts go to row 2 chunk, their im parts to row 2 chunk + 1
    auto image_put = [&](int i, const pre_t &val) __attribute__((always_inline)) {
        const int idx = i * 64 + lane;
        if constexpr (!CPLX) {
            const int r = idx / St::segs, sg = idx % St::segs;
            *reinterpret_cast<pre_t *>(stage + r * St::pitch + sg * St::elems) = val;
        } else {
            const int ch = idx / USEG, u = idx % USEG;
            const IO *e = reinterpret_cast<const IO *>(&val);
            IO *re = stage + (2 * ch) * St::pitch + u * (St::elems / 2), *im = re + St::pitch;
#pragma unroll
            for (int k = 0; k < St::elems / 2; ++k) {
                re[k] = e[2 * k];
                im[k] = e[2 * k + 1];
            }
        }
    };
    auto image_get = [&](int i) __attribute__((always_inline)) -> pre_t {
        const int idx = i * 64 + lane;
        if constexpr (!CPLX) {
            const int r = idx / St::segs, sg = idx % St::segs;
            return *reinterpret_cast<const pre_t *>(stage + r * St::pitch + sg * St::elems);
        } else {
            const int ch = idx / USEG, u = idx % USEG;
            const IO *re = stage + (2 * ch) * St::pitch + u * (St::elems / 2), *im = re + St::pitch;
            pre_t val;
            IO *e = reinterpret_cast<IO *>(&val);
#pragma unroll
            for (int k = 0; k < St::elems / 2; ++k) {
                e[2 * k] = re[k];
                e[2 * k + 1] = im[k];
            }
            return val;
        }
    };
    auto stage_slow = [&](int p) {  // zero beyond the signal
#pragma unroll 1
        for (int i = 0; i < St::per_thread; ++i) {
            const int idx = i * 64 + lane;
            const int r = idx / USEG, sg = idx % USEG;
            // g: index of the unit's first sample (a complex sample for CPLX) in the row
            const int64_t g = (row0 + r) * TI + (int64_t)p * kPiece + (int64_t)sg * (St::elems / LS);
            const int64_t nlim = UP2 ? a.n_in : a.n;
            pre_t val;
            IO *e4 = reinterpret_cast<IO *>(&val);
#pragma unroll
            for (int e = 0; e < St::elems; ++e) e4[e] = (g + e / LS < nlim) ? x[g * LS + e] : IO(0);
            image_put(i, val);
        }
    };

    // .up: output-rate unit (i, lane) of piece p from the input-rate signal: sample m of the row is up * x[m / up] where up divides m
    const int64_t m0 = row0 * T;                                 // output-rate index of the segment's first sample (wave-uniform)
    const int64_t up_q0 = a.up > 1 ? m0 / a.up : 0;
    const unsigned up_r0 = a.up > 1 ? (unsigned)(m0 - up_q0 * a.up) : 0u;
    // A unit holds EL = elems / LS consecutive output-rate samples v .. v + EL - 1; the ones that carry input are the multiples of up: the
    // first at e0 = (up - v mod up) mod up, a second at e0 + up only where that is still inside the unit (EL = 4 with up = 2, 3).  One
    // or two loads and EL selects per unit.  (Before: every element of every unit tested on its own -- three comparisons, a product, a
    // predicated load each: .up cost a third more than the plain filter of the same number of output samples.)
    const IO *xup = x + up_q0 * LS;   // (uniform base + 32-bit lane offsets: the input samples this segment needs start here)
    const int64_t in_left = a.n_in - up_q0;
    const unsigned in_lim = in_left > 0x7fffffff ? 0x7fffffffu : (in_left > 0 ? (unsigned)in_left : 0u);
    // up_request(p): the one or two input samples of every unit of piece p, all requested before any is used (as load_piece does for the plain
    // filter; until round 4 a unit's loads were issued and waited for unit by unit -- 8 dependent round trips per piece, 32 per segment,
    // which is what held rate_change(12).up at 0.13 ms per 2^26 outputs); up_put(p): the units, zero-stuffed, into the image.
    // (they land in pre[][] -- the registers the plain filter's pieces land in: a unit of more than two samples, float32 only, can hold two
    // inputs, anything else one, so a unit's inputs are 16 bytes at most)
    static_assert((St::elems / LS > 2 ? 2 : 1) * LS * sizeof(IO) <= sizeof(pre_t), "a unit's inputs fit its landing register");
    auto up_unit = [&](int p, int i, unsigned &e0, unsigned &qa) __attribute__((always_inline)) {
        constexpr int EL = St::elems / LS;
        const int idx = i * 64 + lane;
        const int r = idx / USEG, sg = idx % USEG;
        // v: the unit's first sample (a complex sample for CPLX) counted from the last multiple of up in front of the segment
        const unsigned v = up_r0 + (unsigned)(r * T + p * kPiece + sg * EL);   // < up + 64 T
        const unsigned q = (unsigned)(((unsigned long long)v * a.up_magic) >> 32);
        const unsigned rem = v - q * (unsigned)a.up;
        e0 = rem ? (unsigned)a.up - rem : 0u;
        qa = q + (rem ? 1u : 0u);
    };
    auto up_request = [&](int p) __attribute__((always_inline)) {
        const unsigned last = in_lim ? in_lim - 1u : 0u;
#pragma unroll
        for (int i = 0; i < St::per_thread; ++i) {
            unsigned e0, qa;
            up_unit(p, i, e0, qa);
            const unsigned q0 = qa < last ? qa : last, q1 = qa + 1u < last ? qa + 1u : last;   // (clamped: the put decides what is used)
            pre_t val = {0.f, 0.f, 0.f, 0.f};
            IO *e4 = reinterpret_cast<IO *>(&val);
#pragma unroll
            for (int c = 0; c < LS; ++c) {
                e4[c] = xup[q0 * LS + c];
                if (St::elems / LS > 2) e4[LS + c] = xup[q1 * LS + c];
            }
            pre[p][i] = val;
        }
    };
    auto up_put = [&](int p) __attribute__((always_inline)) {
        constexpr int EL = St::elems / LS;
        const IO gain = (IO)a.up;
#pragma unroll
        for (int i = 0; i < St::per_thread; ++i) {
            unsigned e0, qa;
            up_unit(p, i, e0, qa);
            const unsigned e1 = e0 + (unsigned)a.up;
            const bool okA = e0 < (unsigned)EL && qa < in_lim, okB = EL > 2 && e1 < (unsigned)EL && qa + 1u < in_lim;
            const pre_t got = pre[p][i];
            const IO *in = reinterpret_cast<const IO *>(&got);
            pre_t val;
            IO *e4 = reinterpret_cast<IO *>(&val);
#pragma unroll
            for (int e = 0; e < St::elems; ++e) {
                const unsigned es = (unsigned)(e / LS);
                const IO va = gain * in[e % LS], vb = EL > 2 ? gain * in[LS + e % LS] : IO(0);
                e4[e] = (okA && es == e0) ? va : ((okB && es == e1) ? vb : IO(0));
            }
            image_put(i, val);
        }
    };
    const bool ld_fast = interior && (a.up == 1 || UP2);
    // .up by 8 or more (from 4 on the unrolled column loop costs more than the zeros: rate_change(4).up 0.130 -> 0.140 ms): at most T / 8 + 1 samples of a chunk are not stuffed zeros, so V = G x is a handful of columns of G per chunk -- formed
    // per lane on the vector ALU from the INPUT samples instead of multiplying the zeros on the matrix pipe (rate_change(12).up: 11 of 128 columns;
    // the chunks of a wave start at different phases of the stuffing, so the columns differ from lane to lane and the matrix form cannot drop them)
    const bool sparse = !DEC && !V32 && a.up >= 8;                         // (.up never comes with a decimating store: none of this in those kernels)
    const int ust = !DEC && (a.up == 2 || a.up == 4) ? a.up : 1;   // .up by 2 / 4: the column step of V = G x (see phase A)

    // ---- A: stream the segment in; chunk rows to registers; V = G x on the matrix pipe --------------------------------
    // (the chunk as 16-byte vectors: as a scalar array hipcc's SROA left half of it in scratch memory)
    typedef IO xv_t __attribute__((ext_vector_type(St::elems)));
    xv_t xq[T / St::elems];
    v4d_t acc[4];
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    v4f_t accf[4];   // V32
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        acc[g] = v4d_t{0.0, 0.0, 0.0, 0.0};
        accf[g] = v4f_t{0.f, 0.f, 0.f, 0.f};
    }
    const int c = lane & 15, j = lane >> 4;
    IO *myrow = stage + lane * St::pitch;
    constexpr int NIN = UPL ? T / 8 : 1;   // UPL: input samples a chunk can hold
    IO xs[NIN];                            // ... this lane's, times the gain, rounded in the signal's type as the staging rounds them
    const int nin = UPL ? T / a.up : 0;    // ... and how many there are
    if constexpr (UPL) {
        const unsigned qa = (unsigned)((lane / LS) * nin);   // the chunk's first input sample, counted from the segment's
        const unsigned last = in_lim ? in_lim - 1u : 0u;
        const IO gain = (IO)a.up;
#pragma unroll
        for (int jj = 0; jj < NIN; ++jj) {
            IO val = IO(0);
            if (jj < nin && in_lim) {
                const unsigned qi = qa + (unsigned)jj;
                const IO got = xup[(qi < last ? qi : last) * LS + lane % LS];
                val = qi < in_lim ? gain * got : IO(0);
            }
            xs[jj] = val;
        }
    }
    xv_t xin[UP2 ? TI / St::elems : 1];   // UP2: the chunk's input samples
    if constexpr (UP2) {
        if (ld_fast) {
#pragma unroll
            for (int p = 0; p < NPI; ++p) load_piece(p);
        }
#pragma unroll
        for (int p = 0; p < NPI; ++p) {
            if (ld_fast) {
#pragma unroll
                for (int i = 0; i < St::per_thread; ++i) image_put(i, pre[p][i]);
            } else {
                stage_slow(p);
            }
            wave_lds_sync();
#pragma unroll
            for (int sgi = 0; sgi < St::segs; ++sgi) xin[p * St::segs + sgi] = *reinterpret_cast<const xv_t *>(myrow + sgi * St::elems);
            // V = G x over every UL-th column of G: input 4 s + j of the piece meets column UL (4 (8 p + s) + j) -- read with a per-lane address, as the
            // zero-stuffed image's .up by 2 / 4 below reads it
            const IO *xu = stage + c * St::pitch + j;
#pragma unroll
            for (int s = 0; s < kPiece / 4; ++s) {
                const int kcol = UL * (4 * (p * (kPiece / 4) + s) + j);
                const double *ga_p = gl + ((kcol >> 2) << 6) + ((kcol & 3) << 4);
                if constexpr (G4) {
                    constexpr int NG = (D + 3) / 4;
                    double ga[NG];
#pragma unroll
                    for (int r = 0; r < NG; ++r) ga[r] = ga_p[4 * r + (lane & 3)];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const double b = (double)xu[g * 16 * St::pitch + 4 * s];
#pragma unroll
                        for (int r = 0; r < NG; ++r) acc[g][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(ga[r], b, acc[g][r], 0, 0, 0);
                    }
                } else if constexpr (V32) {
                    const float gaf = glf[((kcol >> 2) << 6) + ((kcol & 3) << 4) + (lane & 15)];
#pragma unroll
                    for (int g = 0; g < 4; ++g) accf[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(gaf, (float)xu[g * 16 * St::pitch + 4 * s], accf[g], 0, 0, 0);
                } else {
                    const double ga = ga_p[lane & 15];
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga, (double)xu[g * 16 * St::pitch + 4 * s], acc[g], 0, 0, 0);
                }
            }
            wave_lds_sync();
        }
    } else if constexpr (!UPL) {
    // (every piece of the segment is requested up front: the landing registers are the ones the chunk will occupy anyway)
    if (ld_fast) {
#pragma unroll
        for (int p = 0; p < NP; ++p) load_piece(p);
    } else if (a.up > 1) {
        if (in_lim) {
#pragma unroll
            for (int p = 0; p < NP; ++p) up_request(p);
        } else {   // (nothing of the input reaches this segment: up_put selects zeros)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int i = 0; i < St::per_thread; ++i) pre[p][i] = pre_t{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (ld_fast) {
#pragma unroll
            for (int i = 0; i < St::per_thread; ++i) image_put(i, pre[p][i]);
        } else if (a.up > 1) {
            up_put(p);
        } else {
            stage_slow(p);
        }
        wave_lds_sync();
#pragma unroll
        for (int sgi = 0; sgi < St::segs; ++sgi) xq[p * St::segs + sgi] = *reinterpret_cast<const xv_t *>(myrow + sgi * St::elems);
        const IO *xs = stage + c * St::pitch + j;
        if (ust > 1) {
            // .up by 2 or 4: every chunk starts on a multiple of the factor, so the stuffed zeros are the same columns of G in every chunk --
            // the product runs over the other columns only (column ust (4 st + k) of the table for step st, read with a per-lane address)
            const IO *xu = stage + c * St::pitch + j * ust;
            const int nst = (kPiece / 4) / ust;
#pragma unroll
            for (int s = 0; s < kPiece / 8; ++s) {
                if (s < nst) {
                    const int kcol = ust * (4 * (p * nst + s) + j);
                    const double *ga_p = gl + ((kcol >> 2) << 6) + ((kcol & 3) << 4);
                    if constexpr (G4) {
                        constexpr int NG = (D + 3) / 4;
                        double ga[NG];
#pragma unroll
                        for (int r = 0; r < NG; ++r) ga[r] = ga_p[4 * r + (lane & 3)];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const double b = (double)xu[g * 16 * St::pitch + 4 * ust * s];
#pragma unroll
                            for (int r = 0; r < NG; ++r) acc[g][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(ga[r], b, acc[g][r], 0, 0, 0);
                        }
                    } else if constexpr (V32) {
                        const float gaf = glf[((kcol >> 2) << 6) + ((kcol & 3) << 4) + (lane & 15)];
#pragma unroll
                        for (int g = 0; g < 4; ++g) accf[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(gaf, (float)xu[g * 16 * St::pitch + 4 * ust * s], accf[g], 0, 0, 0);
                    } else {
                        const double ga = ga_p[lane & 15];
#pragma unroll
                        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga, (double)xu[g * 16 * St::pitch + 4 * ust * s], acc[g], 0, 0, 0);
                    }
                }
            }
        } else if (!sparse)
#pragma unroll
        for (int s = 0; s < kPiece / 4; ++s) {
            if constexpr (G4) {
                // v_mfma_f64_4x4x4_4b: four independent 4 x 4 x 4 products -- here the SAME four state rows against four groups of four chunks.
                // Its B operand (lane = 16 k + chunk) and its result (lane = 16 (row mod 4) + chunk) lie exactly where the 16 x 16 x 4
                // instruction has them, and register r of the accumulator is row group r; its A operand is G[4 r + (lane & 3)][k = lane >> 4]
                // in every group of four lanes -- read from the same table.  16 cycles against 64 (tools/ubench_mfma_f64_4x4.hip: 7.3 / 35 ns),
                // and only the row groups that hold states are multiplied: 4 biquads pay 2 x 16 cycles per step and column tile, not 64.
                constexpr int NG = (D + 3) / 4;   // row groups that hold states (rows >= D of the table are zero)
                double ga[NG];
#pragma unroll
                for (int r = 0; r < NG; ++r) ga[r] = gl[(p * (kPiece / 4) + s) * 64 + (lane & 48) + 4 * r + (lane & 3)];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double b = (double)xs[g * 16 * St::pitch + 4 * s];
#pragma unroll
                    for (int r = 0; r < NG; ++r) acc[g][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(ga[r], b, acc[g][r], 0, 0, 0);
                }
            } else if constexpr (V32) {
                const float gaf = glf[(p * (kPiece / 4) + s) * 64 + lane];
#pragma unroll
                for (int g = 0; g < 4; ++g) accf[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(gaf, (float)xs[g * 16 * St::pitch + 4 * s], accf[g], 0, 0, 0);
            } else {
                const double ga = gl[(p * (kPiece / 4) + s) * 64 + lane];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga, (double)xs[g * 16 * St::pitch + 4 * s], acc[g], 0, 0, 0);
            }
        }
        wave_lds_sync();
    }
    }

    // chunk end states from the accumulator layout (column = lane & 15, state row = (lane >> 4) + 4 reg) to one lane per chunk
    double v[D];
    if constexpr (UPL) {
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = 0.0;
#pragma unroll
        for (int jj = 0; jj < NIN; ++jj) {
            if (jj < nin) {
                const int kk = jj * a.up;                                      // column of G: the same for every chunk
                const double *gp = gtab + (kk >> 2) * 64 + 16 * (kk & 3);     // (its D state rows are consecutive in the operand order of the table)
                const double xv = (double)xs[jj];
#pragma unroll
                for (int d = 0; d < D; ++d) v[d] = fma(gp[d], xv, v[d]);
            }
        }
#pragma unroll
        for (int k = 0; k < NSEC; ++k) *reinterpret_cast<v2d_t *>(E + (k * 64 + lane) * 2) = v2d_t{v[2 * k], v[2 * k + 1]};
        wave_lds_sync();
    } else if (sparse) {
        constexpr int JMAX = T / 8 + 1;
        const unsigned v0 = up_r0 + (unsigned)((lane / LS) * T);          // this lane's chunk start, counted from the last multiple of up in front of the segment
        const unsigned q = (unsigned)(((unsigned long long)v0 * a.up_magic) >> 32);
        const unsigned rem = v0 - q * (unsigned)a.up;
        const unsigned qa = q + (rem ? 1u : 0u), k0 = rem ? (unsigned)a.up - rem : 0u;   // first input sample of the chunk and its place in it
        const unsigned last = in_lim ? in_lim - 1u : 0u;
        const IO gain = (IO)a.up;
        IO xin[JMAX];
#pragma unroll
        for (int jj = 0; jj < JMAX; ++jj) {
            const unsigned qi = qa + (unsigned)jj;
            xin[jj] = in_lim ? xup[(qi < last ? qi : last) * LS + lane % LS] : IO(0);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = 0.0;
#pragma unroll
        for (int jj = 0; jj < JMAX; ++jj) {
            const unsigned k = k0 + (unsigned)jj * (unsigned)a.up;
            const bool ok = k < (unsigned)T && qa + (unsigned)jj < in_lim;
            const double xv = ok ? (double)(gain * xin[jj]) : 0.0;             // (the value the staging put into the chunk: rounded in the signal's type)
            const unsigned kk = k < (unsigned)T ? k : (unsigned)(T - 1);
            const double *gp = gl + (kk >> 2) * 64 + 16 * (kk & 3);             // column kk of G: its D state rows are consecutive
#pragma unroll
            for (int d = 0; d < D; d += 2) {
                const v2d_t g2 = *reinterpret_cast<const v2d_t *>(gp + d);
                v[d] = fma(g2[0], xv, v[d]);
                v[d + 1] = fma(g2[1], xv, v[d + 1]);
            }
        }
#pragma unroll
        for (int k = 0; k < NSEC; ++k) *reinterpret_cast<v2d_t *>(E + (k * 64 + lane) * 2) = v2d_t{v[2 * k], v[2 * k + 1]};
        wave_lds_sync();
    } else {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // state row: section d >> 1, component d & 1.  The FP64 instruction leaves row (lane >> 4) + 4 reg in register reg, the float32 one row 4 (lane >> 4) + reg
            const int d = V32 ? 4 * j + r : j + 4 * r;
            if (V32 || 4 * r < D) {
                const double av = V32 ? (double)accf[g][r] : acc[g][r];
                if (d < D) E[(((d >> 1) * 64) + 16 * g + c) * 2 + (d & 1)] = UP2 ? (double)UL * av : av;   // (UP2: the gain the zero-stuffed image would have carried)
            }
        }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < NSEC; ++k) {
        const v2d_t t = *reinterpret_cast<const v2d_t *>(E + (k * 64 + lane) * 2);
        v[2 * k] = t[0];
        v[2 * k + 1] = t[1];
    }
    }

    // ---- S: from-rest inclusive scan of the wave's 64 chunk states ------------------------------------------------------
#pragma unroll 1
    for (int l = 0; l < a.n_lv; ++l) {
        const int s = LS << l;
        const int src = lane >= s ? lane - s : lane;
        double left[D];
#pragma unroll
        for (int k = 0; k < NSEC; ++k) {
            const v2d_t t = *reinterpret_cast<const v2d_t *>(E + (k * 64 + src) * 2);
            left[2 * k] = t[0];
            left[2 * k + 1] = t[1];
        }
        wave_lds_sync();
        if (lane >= s) blocks_acc<NSEC>(lvl + (size_t)l * NSEC * 4, left, v);
#pragma unroll
        for (int k = 0; k < NSEC; ++k) *reinterpret_cast<v2d_t *>(E + (k * 64 + lane) * 2) = v2d_t{v[2 * k], v[2 * k + 1]};
        wave_lds_sync();
    }
    double z[D];
#pragma unroll
    for (int k = 0; k < NSEC; ++k) {
        const v2d_t t = *reinterpret_cast<const v2d_t *>(E + (k * 64 + (lane >= LS ? lane - LS : 0)) * 2);
        z[2 * k] = lane >= LS ? t[0] : 0.0;
        z[2 * k + 1] = lane >= LS ? t[1] : 0.0;
    }

    // ---- L: publish the segment's end state(s) from rest, fetch the K predecessors' ------------------------------------
    // (CPLX: two states per segment -- lanes 62 / 63 hold the re / im stream's; granule g belongs to stream g >> 5)
    unsigned *cw = reinterpret_cast<unsigned *>(lds_raw + wave * kWaveStage + 64 * 16 * 8);   // behind the scan exchange, inside the (idle) image
    if (lane < GR && (lane & 31) < 2 * D) {
        const int d = (lane & 31) >> 1, src_lane = CPLX ? 62 + (lane >> 5) : 63;
        const unsigned half = reinterpret_cast<const unsigned *>(E)[((((d >> 1) * 64) + src_lane) * 2 + (d & 1)) * 2 + (lane & 1)];
        __hip_atomic_store(a.lb + (size_t)tk * GR + lane, ((unsigned long long)a.epoch << 32) | half, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (seg > 0) {
#pragma unroll 1
        for (int k0 = 0; k0 < a.K; k0 += 64 / GR) {
            const int back = k0 + (CPLX ? 0 : (lane >> 5)) + 1;   // this lane's predecessor distance
            const int gi = CPLX ? lane : (lane & 31);            // its granule
            unsigned got = 0;
            if (back <= a.K && back <= seg && (gi & 31) < 2 * D) {
                const unsigned long long *srcp = a.lb + (size_t)(tk - back) * GR + gi;
                unsigned long long g = 0;
                int spins = 0;
                for (;;) {
                    g = __hip_atomic_load(srcp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(g >> 32) == a.epoch) break;
                    if (++spins > (1 << 22)) {  // ~ seconds: never in a healthy run; fail loudly instead of hanging the GPU
                        *a.err = 1u;
                        g = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                got = (unsigned)g;
            }
            if (back <= KMAX) cw[(back - 1) * GR + gi] = got;
        }
        wave_lds_sync();

        // ---- C: z_j += Phi^j c,  c = sum_m Psi^m P_(s-1-m) ---------------------------------------------------------------
        double u[D];
        const double *cwd = reinterpret_cast<const double *>(cw) + (CPLX ? (lane & 1) * 16 : 0);   // (this lane's stream)
        const int cj = lane / LS;   // chunk index inside the segment
#pragma unroll
        for (int k = 0; k < NSEC; ++k) {
            const v2d_t t = *reinterpret_cast<const v2d_t *>(cwd + 2 * k);
            u[2 * k] = t[0];
            u[2 * k + 1] = t[1];
        }
#pragma unroll 1
        for (int m = 1; m < a.K; ++m) {
            double pm[D];
#pragma unroll
            for (int k = 0; k < NSEC; ++k) {
                const v2d_t t = *reinterpret_cast<const v2d_t *>(cwd + m * (GR / 2) + 2 * k);
                pm[2 * k] = t[0];
                pm[2 * k + 1] = t[1];
            }
            blocks_acc<NSEC>(psi + (size_t)(m - 1) * NSEC * 4, pm, u);
        }
        // (Phi^j by the binary digits of j: up to six rounds of 2 x 2 products, every lane through every level some lane needs -- ~480 vector instructions per
        // segment for the 32 multiply-adds a lane wants.  Round 6 tried the lane's own row of a [64][NSEC][4] table of Phi^j instead, requested in front of the
        // look-back's poll: 64 more live registers next to the chunk, 180 bytes of scratch per lane in the plain 8-biquad kernel; not kept.)
#pragma unroll 1
        for (int l = 0; l < a.n_lv; ++l) {
            if ((cj >> l) & 1) {
                double t2[D];
#pragma unroll
                for (int d = 0; d < D; ++d) t2[d] = 0.0;
                blocks_acc<NSEC>(lvl + (size_t)l * NSEC * 4, u, t2);
#pragma unroll
                for (int d = 0; d < D; ++d) u[d] = t2[d];
            }
        }
        if ((cj >> a.n_lv) == 0) {   // (Phi^j c is negligible for j >= 2^n_lv)
#pragma unroll
            for (int d = 0; d < D; ++d) z[d] += u[d];
        }
    }
    wave_lds_sync();  // the image is free again
    if (PRIO) __builtin_amdgcn_s_setprio(0);

    // ---- B: the recurrence over the register-resident chunk; outputs leave through the LDS image ------------------------
    // (the output taps live in VGPRs: 33 double coefficients next to the addresses do not fit a wave's 102 SGPRs, and
    // hipcc then shuffles them through v_readlane / re-reads them from the kernel arguments inside the body)
    double al[NSEC], be[NSEC], gam = cf.gamma;
    if constexpr (!UPL) {   // (UPL: the taps are row 0 of its table, and that arrives by scalar loads)
#pragma unroll
        for (int s = 0; s < NSEC; ++s) {
            al[s] = cf.al[s];
            be[s] = cf.be[s];
            asm volatile("" : "+v"(be[s]));
        }
    }
    // .dn, compact form: this lane's sample stream (T samples of chunk cj, component lane % LS) walks the positions
    // v = R0 + cj T + c of the segment, R0 = (first sample of the segment) mod dec; sample v is kept iff dec divides v, as
    // output Q0 + v / dec of the row.  (dq, dt) = (v / dec, v mod dec) of the next 16-byte unit, advanced by its size per unit.
    const bool compact = DEC && a.dec_compact;
    constexpr bool rounds = DECM == 2;
    int64_t dec_q0 = 0;
    unsigned dec_r0 = 0, dq = 0, dt = 0;
    if (DEC) {
        dec_q0 = m0 / a.dec;
        dec_r0 = (unsigned)(m0 - dec_q0 * a.dec);
        const unsigned v0 = dec_r0 + (unsigned)((lane / LS) * T);
        dq = (unsigned)(((unsigned long long)v0 * a.dec_magic) >> 32);
        dt = v0 - dq * (unsigned)a.dec;
    }
    const unsigned dec_ob = dec_r0 != 0 ? 1u : 0u;     // first output of the segment, relative to dec_q0
    const int64_t n_out = a.n_keep / a.dec;
    const int64_t olim64 = n_out - dec_q0;
    const unsigned olim = olim64 <= 0 ? 0u : (olim64 > 0x7fffffff ? 0x7fffffffu : (unsigned)olim64);   // outputs of this row from dec_q0 on
    // .dn, compact form: an output is a 2 NSEC + 1 term sum over the states, and only every dec-th one is kept -- the sum is formed for the
    // kept samples only (e0r: the kept element of the current 16-byte unit, if any, as the gathering below finds it).  The lanes of a wave
    // walk different phases unless dec divides the chunk length, so a sample's sum is skipped only where NO lane keeps it: at dec = 12 the
    // 64 chunks of a segment start at three different phases (128 mod 12 = 8) and three of twelve samples pay for the sum instead of all.
    unsigned dtr = dt, e0r = 0;
    int upj_cnt = 0;   // UPJ: (samples since the last input sample) - 1
    constexpr bool UPJ_PRE = UPJ && NSEC <= 4;
    // UPJ, up to 4 biquads: the rows c A^cnt arrive by scalar loads, wave-uniform, in PAIRS (rows cnt, cnt + 1 are adjacent in the table, row `up` repeats
    // row 0), requested a pair ahead.  Scalar loads return in no order, so a wave can only wait for ALL of them: with one row requested per sample the wait
    // for this sample's row was also the wait for the next one's, one sample old -- 200 clocks of scalar-cache latency against 40 of arithmetic, and four
    // waves per SIMD do not cover that (SQ_WAIT_ANY: half of every wave's life).  Pairs halve the waits and double the distance.
    double cjn[UPJ_PRE ? 2 * D : 1];   // the pair requested last: rows of the next two samples once the current pair is used up
    double cjc[UPJ_PRE ? 2 * D : 1];   // the pair in use
    int upj_c2 = 2;                    // first row of the pair to request next
    // more than 4 biquads: a row is 2 NSEC doubles = up to 32 scalar registers, a pair in use and a pair in flight would be all there are: one row, a sample ahead
    double cj1[(UPJ && !UPJ_PRE) ? D : 1];
    if constexpr (UPJ && !UPJ_PRE) {
#pragma unroll
        for (int d = 0; d < D; ++d) cj1[d] = upj[d];
    }
    if constexpr (UPJ_PRE) {
#pragma unroll
        for (int d = 0; d < 2 * D; ++d) cjn[d] = upj[d];
#pragma unroll
        for (int d = 0; d < 2 * D; ++d) cjc[d] = 0.0;
    }
    if constexpr (UPL) {
        // (the blocks of A^up, requested at the top of the kernel, have arrived: said HERE, so that no wait for them -- which would also be a wait for the
        // wave's own stores -- appears at the input samples inside the loop)
#pragma unroll
        for (int i = 0; i < 4 * NSEC; ++i) asm volatile("" : "+v"(AuV[i]));
    }
    int dnl_cnt = 0;                                  // DNL: samples since the last kept one
    unsigned dnl_o = (unsigned)(lane / LS) * (unsigned)(T / (DNL ? a.dec : 1));   // DNL: this lane's next output, counted from the segment's first
    int upl_ji = 1;                                   // UPL: the next input sample of the chunk
    double upl_next = UPL ? (double)xs[NIN > 1 ? 1 : 0] : 0.0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if constexpr (UPL || UP2) wave_lds_sync();   // (the wave's rows are free: UPL / UP2 hand every 16-byte unit to the image as it completes)
        xv_t ou = xv_t(0);   // UP2: the unit being completed
#pragma unroll
        for (int k = 0; k < kPiece; ++k) {
            constexpr int kE = St::elems;
            const int e = (p * kPiece + k) % kE;
            if (DEC && !UNI && e == 0) {
                e0r = dtr == 0 ? 0u : (unsigned)a.dec - dtr;
                dtr += kE;
                dtr = dtr >= (unsigned)a.dec ? dtr - (unsigned)a.dec : dtr;
                dtr = dtr >= (unsigned)a.dec ? dtr - (unsigned)a.dec : dtr;   // (M = 2, 3: a 4-sample unit spans more than one period)
            }
            double xd;
            const bool zero_in = UP2 && ((p * kPiece + k) % UL) != 0;   // UP2: a stuffed zero (known here, not tested)
            if constexpr (UPL) xd = p * kPiece + k == 0 ? (double)xs[0] : upl_next;   // (read at input samples only)
            else if constexpr (UP2) xd = zero_in ? 0.0 : (double)UL * (double)xin[((p * kPiece + k) / UL) / kE][((p * kPiece + k) / UL) % kE];
            else xd = (double)xq[(p * kPiece + k) / kE][e];
            if constexpr (UPJ) {
                if (p * kPiece + k > 0) {   // (the chunk's first sample is an input sample met with the state from the scan: the plain step below)
                    // (up to 4 biquads the row c A^cnt of THIS sample was requested a sample ago -- scalar loads, wave-uniform -- and the next one goes
                    // out first; longer rows would not leave the scalar registers for two of them)
                    double cj[D];
                    const bool is_input = upj_cnt == a.up - 1;
                    if constexpr (UPJ_PRE) {
                        const int half = (p * kPiece + k - 1) & 1;   // samples 1, 2 are the first pair
                        if (half == 0) {
#pragma unroll
                            for (int d = 0; d < 2 * D; ++d) cjc[d] = cjn[d];
                            asm volatile("" : "+s"(upj_c2));   // (one address at a time: hipcc otherwise works out the addresses of many pairs ahead and spills them)
                            const double *nr = upj + (size_t)upj_c2 * D;
#pragma unroll
                            for (int d = 0; d < 2 * D; ++d) cjn[d] = nr[d];
                            upj_c2 += 2;                                   // (rows c2, c2 + 1; the pair behind row up - 1 or up starts over)
                            upj_c2 = upj_c2 >= a.up ? upj_c2 - a.up : upj_c2;
                        }
#pragma unroll
                        for (int d = 0; d < D; ++d) cj[d] = cjc[half * D + d];
                        upj_cnt = is_input ? 0 : upj_cnt + 1;
                    } else {
#pragma unroll
                        for (int d = 0; d < D; ++d) cj[d] = cj1[d];
                        upj_cnt = is_input ? 0 : upj_cnt + 1;
                        const double *nr = upj + (size_t)upj_cnt * D;
#pragma unroll
                        for (int d = 0; d < D; ++d) cj1[d] = nr[d];
                    }
                    double yv = cj[0] * z[0];
                    yv = fma(cj[1], z[1], yv);
#pragma unroll
                    for (int s = 1; s < NSEC; ++s) {
                        yv = fma(cj[2 * s], z[2 * s], yv);
                        yv = fma(cj[2 * s + 1], z[2 * s + 1], yv);
                    }
                    if (is_input) {   // an input sample: its direct term, and the state jumps to right behind it (the blocks of A^up: one sample in `up`)
                        yv = fma(gam, xd, yv);
                        const double *Au = AuV;
#pragma unroll
                        for (int s = 0; s < NSEC; ++s) {
                            const double n0 = fma(Au[4 * s + 1], z[2 * s + 1], fma(Au[4 * s], z[2 * s], xd));
                            const double n1 = fma(Au[4 * s + 3], z[2 * s + 1], Au[4 * s + 2] * z[2 * s]);
                            z[2 * s] = n0;
                            z[2 * s + 1] = n1;
                        }
                        if constexpr (UPL) {
                            upl_ji = upl_ji + 1 < NIN ? upl_ji + 1 : NIN - 1;
                            upl_next = (double)xs[upl_ji];   // (a wave-uniform index into registers)
                        }
                    }
                    xq[(p * kPiece + k) / kE][e] = (IO)yv;
                    if constexpr (UPL) {
                        if (e == kE - 1) *reinterpret_cast<xv_t *>(myrow + (k / kE) * kE) = xq[(p * kPiece + k) / kE];
                    }
                    continue;
                }
                if constexpr (UPL) {   // the chunk's first sample: an input sample met with the state from the scan -- the taps are row 0, the step the plain one
                    double yv = gam * xd;
#pragma unroll
                    for (int d = 0; d < D; ++d) yv = fma(UPJ_PRE ? cjn[d] : cj1[d], z[d], yv);   // (the first pair / row, requested above, is or starts with row 0)
                    xq[0][0] = (IO)yv;
#pragma unroll
                    for (int s = 0; s < NSEC; ++s) {
                        const double w0 = fma(cf.na2[s], z[2 * s + 1], fma(cf.na1[s], z[2 * s], xd));
                        z[2 * s + 1] = z[2 * s];
                        z[2 * s] = w0;
                    }
                    continue;
                }
            }
            bool keep;
            if constexpr (UNI) keep = dnl_cnt == 0;
            else keep = !DEC || e0r == (unsigned)e || e0r + (unsigned)a.dec == (unsigned)e;   // (the unit's kept samples: e0r and, with M below the samples of a unit, e0r + M)
            if constexpr (UNI) dnl_cnt = dnl_cnt + 1 == a.dec ? 0 : dnl_cnt + 1;
            if constexpr (DNL) {
                if (keep) {
                    double yv = gam * xd;
#pragma unroll
                    for (int s = 0; s < NSEC; ++s) {
                        yv = fma(al[s], z[2 * s], yv);
                        yv = fma(be[s], z[2 * s + 1], yv);
                    }
                    if (dnl_o < olim) {
                        const unsigned slot = dnl_o * LS + (unsigned)(lane % LS);
                        stage[slot + (slot >> 5)] = (IO)yv;
                    }
                    ++dnl_o;
                }
            } else if (keep) {
                // (one chain of 2 NSEC + 1 dependent operations per output.  Measured and not kept, round 6: two chains -- even and odd sections -- and one add:
                // config 4 0.1435 -> 0.1455 ms, .dn(x, 3) 0.136 -> 0.140; the recurrences of the same sample already fill the chain's gaps)
                double yv = zero_in ? al[0] * z[0] : fma(al[0], z[0], gam * xd);
                yv = fma(be[0], z[1], yv);
#pragma unroll
                for (int s = 1; s < NSEC; ++s) {
                    yv = fma(al[s], z[2 * s], yv);
                    yv = fma(be[s], z[2 * s + 1], yv);
                }
                if constexpr (UP2) {
                    ou[e] = (IO)yv;
                    if (e == kE - 1) *reinterpret_cast<xv_t *>(myrow + (k / kE) * kE) = ou;
                } else {
                    xq[(p * kPiece + k) / kE][e] = (IO)yv;
                }
            }
#pragma unroll
            for (int s = 0; s < NSEC; ++s) {
                const double w0 = fma(cf.na2[s], z[2 * s + 1], zero_in ? cf.na1[s] * z[2 * s] : fma(cf.na1[s], z[2 * s], xd));
                z[2 * s + 1] = z[2 * s];
                z[2 * s] = w0;
            }
        }
        if constexpr (DNL) continue;   // (the kept outputs are in the image already)
        if (compact) {
#pragma unroll
            for (int sgi = 0; sgi < St::segs; ++sgi) {
                const xv_t u = xq[p * St::segs + sgi];
                const unsigned e0 = dt == 0 ? 0u : (unsigned)a.dec - dt;    // the unit's kept sample, if < elems (dec >= elems: at most one)
                IO pick = u[0];
#pragma unroll
                for (int e = 1; e < St::elems; ++e) pick = (e0 == (unsigned)e) ? u[e] : pick;
                const unsigned oq = dq + (dt != 0);                          // its output, relative to dec_q0
                if (e0 < (unsigned)St::elems && oq < olim) {
                    const unsigned slot = (oq - dec_ob) * LS + (unsigned)(lane % LS);
                    stage[slot + (slot >> 5)] = pick;                        // (one pad per 32: the lanes' runs start T / dec slots apart)
                }
                dt += St::elems;
                if (dt >= (unsigned)a.dec) { dt -= (unsigned)a.dec; dq += 1; }
            }
            continue;
        }
        if (rounds) continue;   // (the outputs stay in the registers of the chunk until the gathering below)
        if (PRIO && SK_PAR_PRIO_ST) __builtin_amdgcn_s_setprio(SK_PAR_PRIO);   // (the piece's way out through the image: -2.7 %)
        if constexpr (!UPL && !UP2) {
            wave_lds_sync();  // the wave's rows are free (its previous piece's stores have read them)
#pragma unroll
            for (int sgi = 0; sgi < St::segs; ++sgi) *reinterpret_cast<xv_t *>(myrow + sgi * St::elems) = xq[p * St::segs + sgi];
        }
        wave_lds_sync();
        if (!DEC && interior) {
            {
                // (all reads of the image first, into the registers the piece has just left: issued one behind the other they cost
                // one LDS round trip; hipcc otherwise cycles two staging vectors through four read -> wait -> store rounds)
                pre_t outq[St::per_thread];
#pragma unroll
                for (int i = 0; i < St::per_thread; ++i) outq[i] = image_get(i);
#pragma unroll
                for (int i = 0; i < St::per_thread; ++i) asm volatile("" : "+v"(outq[i]));
#pragma unroll
                for (int i = 0; i < St::per_thread; ++i)
                    __builtin_nontemporal_store(outq[i], reinterpret_cast<pre_t *>(yseg + (loff + i * kRowStep + p * kPiece * LS)));
            }
            if (PRIO && SK_PAR_PRIO_ST) __builtin_amdgcn_s_setprio(0);
            continue;
        }
        int64_t dq_run = 0;
        int dr_run = 0;
#pragma unroll 1
        for (int i = 0; i < St::per_thread; ++i) {
            const int idx = i * 64 + lane;
            const int r = idx / USEG, sg = idx % USEG;
            const int64_t g = (row0 + r) * T + (int64_t)p * kPiece + (int64_t)sg * (St::elems / LS);   // first sample of the unit
            const pre_t val = image_get(i);
            const IO *tmp = reinterpret_cast<const IO *>(&val);
            if (DEC && !CPLX) {
                // decimating store (as in iir_fused_kernel): segment i of this lane starts a fixed number of samples after
                // segment i - 1, so its (quotient, remainder) by dec follow from the first by adding (dec_dq, dec_dr)
                if (i == 0) {
                    dq_run = g / a.dec;
                    dr_run = (int)(g - dq_run * a.dec);
                }
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
            } else if (g < a.n) {
#pragma unroll
                for (int e = 0; e < St::elems; ++e)
                    if (g + e / LS < a.n) y[g * LS + e] = tmp[e];
            }
        }
    }
    if constexpr (rounds) {
        // dec = 2, 3 on 4-sample units: the chunks of the segment in a.dec_rounds ranges; a range's kept outputs are a run of the row
        const unsigned M = (unsigned)a.dec;
        const int per = CH / a.dec_rounds;                              // chunks per range
        const int cj = lane / LS;
        const unsigned v0 = dec_r0 + (unsigned)(cj * T);
        const unsigned dq0 = (unsigned)(((unsigned long long)v0 * a.dec_magic) >> 32), dt0 = v0 - dq0 * M;
#pragma unroll 1
        for (int rd = 0; rd < a.dec_rounds; ++rd) {
            const unsigned vb = dec_r0 + (unsigned)(rd * per * T);       // the range's first sample
            const unsigned ob_r = (vb + M - 1) / M;                        // its first output, relative to dec_q0
            if (cj / per == rd) {
                unsigned q = dq0, t = dt0;
#pragma unroll
                for (int ui = 0; ui < T / St::elems; ++ui) {
                    const xv_t u = xq[ui];
                    // the unit's kept samples: e0 = (-t) mod M and, if it still lies inside the unit, e0 + M
                    const unsigned e0 = t == 0 ? 0u : M - t, e1 = e0 + M;
                    IO p0 = u[0], p1 = u[St::elems - 1];
#pragma unroll
                    for (int e = 1; e < St::elems; ++e) {
                        p0 = (e0 == (unsigned)e) ? u[e] : p0;
                        p1 = (e1 == (unsigned)e) ? u[e] : p1;
                    }
                    const unsigned o0 = q + (t != 0);                      // output of the sample at e0, relative to dec_q0
                    const unsigned slot = (o0 - ob_r) * LS + (unsigned)(lane % LS);
                    if (e0 < (unsigned)St::elems && o0 < olim) stage[slot + (slot >> 5)] = p0;
                    if (e1 < (unsigned)St::elems && o0 + 1 < olim) stage[slot + LS + ((slot + LS) >> 5)] = p1;
                    t += St::elems;
                    const unsigned kk = (t >= M) + (t >= 2 * M);
                    t -= kk * M;
                    q += kk;
                }
            }
            wave_lds_sync();
            const unsigned ve = dec_r0 + (unsigned)((rd + 1) * per * T);
            unsigned oe_r = (ve + M - 1) / M;
            if (oe_r > olim) oe_r = olim;
            const int cnt = oe_r > ob_r ? (int)(oe_r - ob_r) * LS : 0;
            IO *yo = y + (dec_q0 + ob_r) * LS;
#pragma unroll 1
            for (int i = lane; i < cnt; i += 64) __builtin_nontemporal_store(stage[i + (i >> 5)], yo + i);
            wave_lds_sync();
        }
    }
    if (compact) {
        // the segment's kept outputs: one contiguous run of the row, y[dec_q0 + dec_ob ...)
        wave_lds_sync();
        const int64_t ob = dec_q0 + dec_ob;
        int64_t oe = dec_q0 + (int64_t)((dec_r0 + (unsigned)(CH * T) + (unsigned)a.dec - 1) / (unsigned)a.dec);
        if (oe > n_out) oe = n_out;
        const int cnt = oe > ob ? (int)(oe - ob) * LS : 0;
        IO *yo = y + ob * LS;
#pragma unroll 1
        for (int i = lane; i < cnt; i += 64) __builtin_nontemporal_store(stage[i + (i >> 5)], yo + i);
    }
}

// ------------------------------------------------------------------------------------------------------- host side
struct ParTables {
    int T = 0;
    int n_lv = 0, K = 0;         // K = 0: the filter remembers more than kParMaxK segments of this length (not served)
    double *gt_dev = nullptr;    // G in MFMA A-operand order [T / 4][64]
    double *lvl_dev = nullptr;   // Phi^(2^l), l = 0..5: [6][nsec][4]
    double *psi_dev = nullptr;   // Psi^m, m = 1..kParMaxK-1, Psi = Phi^(chunks per segment): [kParMaxK - 1][nsec][4]
    int v32 = 0;                 // V = G x in float32 for chunks of this length (par_v32_probe): 0 not probed, 1 admitted, -1 refused
    double v32_err = 0.0;        // ... the worst probe error it showed
};

struct ParPlan {
    int state = 0;               // 0 untested, 1 expansion accepted, -1 not applicable
    int nsec = 0;
    long double a1[8], a2[8], r0[8], r1[8], c0 = 0.0L;
    double na1[8], na2[8], al[8], be[8], gamma = 0.0;
    double kappa = 0.0, ir_err = 0.0, l1h = 0.0;   // (l1h: the l1 norm of the impulse response -- the forward bound per unit input)
    ParTables tab[8];            // [0] float32 (T = 128), [1] float64 (T = 64), [2] complex64, [3] complex128 (32 chunks per segment), [4] / [5] float32 / complex64 with T = 96 (.dn, .up), [6] / [7] float64 / complex128 with T = 96 (.up)
    struct UpJump { int up; double *dev; };
    std::vector<UpJump> upj;     // per L: [L][2 nsec] rows c A^j + [nsec][4] blocks of A^L (UPJ kernels)
    unsigned long long *lbg_dev = nullptr;
    size_t lbg_cap = 0;
    unsigned long long *ticket_dev = nullptr;
    unsigned long long ticket_count = 0;
    unsigned epoch = 0;
};

void iir_par_free(ParPlan *p)
{
    if (!p) return;
    for (ParTables &t : p->tab) {
        if (t.gt_dev) (void)hipFree(t.gt_dev);
        if (t.lvl_dev) (void)hipFree(t.lvl_dev);
        if (t.psi_dev) (void)hipFree(t.psi_dev);
    }
    for (auto &u : p->upj) if (u.dev) (void)hipFree(u.dev);
    if (p->lbg_dev) (void)hipFree(p->lbg_dev);
    if (p->ticket_dev) (void)hipFree(p->ticket_dev);
    delete p;
}

namespace {
struct Q2 { long double u, v; };   // u + v q  in R[q] / (1 + a1 q + a2 q^2)

// partial fractions of prod_k B_k(q) / A_k(q), q = z^-1, by arithmetic modulo each denominator
bool par_expand(const double *coef, int nsec, ParPlan &P)
{
    long double b[8][3], a[8][3];
    int degA[8], degB[8], sumA = 0, sumB = 0;
    for (int k = 0; k < nsec; ++k) {
        const double *c = coef + 5 * k;
        b[k][0] = c[0]; b[k][1] = c[1]; b[k][2] = c[2];
        a[k][0] = 1.0L; a[k][1] = c[3]; a[k][2] = c[4];
        for (int i = 0; i < 5; ++i)
            if (!std::isfinite(c[i])) return false;
        degA[k] = a[k][2] != 0.0L ? 2 : (a[k][1] != 0.0L ? 1 : 0);
        degB[k] = b[k][2] != 0.0L ? 2 : (b[k][1] != 0.0L ? 1 : 0);
        sumA += degA[k];
        sumB += degB[k];
    }
    if (sumB > sumA) return false;   // a polynomial part beyond the direct term: not a sum of these branches
    P.c0 = 0.0L;
    if (sumB == sumA) {
        P.c0 = 1.0L;
        for (int k = 0; k < nsec; ++k) P.c0 *= b[k][degB[k]] / a[k][degA[k]];
    }
    for (int k = 0; k < nsec; ++k) {
        P.a1[k] = a[k][1];
        P.a2[k] = a[k][2];
        P.r0[k] = P.r1[k] = 0.0L;
        if (degA[k] == 2) {
            const long double a1 = a[k][1], a2 = a[k][2];
            auto red = [&](const long double *c) { return Q2{c[0] - c[2] / a2, c[1] - c[2] * a1 / a2}; };
            auto mul = [&](Q2 x, Q2 y) {
                const long double vv = x.v * y.v;
                return Q2{x.u * y.u - vv / a2, x.u * y.v + x.v * y.u - vv * a1 / a2};
            };
            Q2 acc{1.0L, 0.0L};
            for (int jx = 0; jx < nsec; ++jx) {
                acc = mul(acc, red(b[jx]));
                if (jx == k) continue;
                const Q2 d = red(a[jx]);
                // inverse of d: [u, -v/a2; v, u - v a1/a2] [s; t] = [1; 0]
                const long double m11 = d.u, m12 = -d.v / a2, m21 = d.v, m22 = d.u - d.v * a1 / a2;
                const long double det = m11 * m22 - m12 * m21;
                const long double scale = fabsl(m11 * m22) + fabsl(m12 * m21);
                if (!(fabsl(det) > 1e-12L * scale) || !std::isfinite((double)det)) return false;   // a pole shared with another section
                acc = mul(acc, Q2{m22 / det, -m21 / det});
            }
            P.r0[k] = acc.u;
            P.r1[k] = acc.v;
        } else if (degA[k] == 1) {
            const long double q0 = -1.0L / a[k][1];
            long double val = 1.0L;
            for (int jx = 0; jx < nsec; ++jx) {
                val *= b[jx][0] + b[jx][1] * q0 + b[jx][2] * q0 * q0;
                if (jx == k) continue;
                const long double den = a[jx][0] + a[jx][1] * q0 + a[jx][2] * q0 * q0;
                if (!(fabsl(den) > 1e-12L)) return false;
                val /= den;
            }
            P.r0[k] = val;
        }
        if (!std::isfinite((double)P.r0[k]) || !std::isfinite((double)P.r1[k])) return false;
    }
    if (!std::isfinite((double)P.c0)) return false;
    long double gam = P.c0;
    for (int k = 0; k < nsec; ++k) {
        P.na1[k] = (double)(-P.a1[k]);
        P.na2[k] = (double)(-P.a2[k]);
        P.al[k] = (double)(P.r1[k] - P.r0[k] * P.a1[k]);
        P.be[k] = (double)(-P.r0[k] * P.a2[k]);
        gam += P.r0[k];
    }
    P.gamma = (double)gam;
    // acceptance: the expansion with its double coefficients against the cascade (long double DF2T), impulse response
    const int NI = 8192;
    long double zc[16] = {0}, w1[8] = {0}, w2[8] = {0};
    long double hmax = 0.0L, emax = 0.0L, l1h = 0.0L, l1b = fabsl((long double)P.gamma);
    for (int n = 0; n < NI; ++n) {
        long double xin = n == 0 ? 1.0L : 0.0L, xc = xin;
        for (int s = 0; s < nsec; ++s) {
            const double *c = coef + 5 * s;
            const long double yv = (long double)c[0] * xc + zc[2 * s];
            zc[2 * s] = (long double)c[1] * xc - (long double)c[3] * yv + zc[2 * s + 1];
            zc[2 * s + 1] = (long double)c[2] * xc - (long double)c[4] * yv;
            xc = yv;
        }
        long double yp = (long double)P.gamma * xin;
        for (int s = 0; s < nsec; ++s) {
            const long double br = (long double)P.al[s] * w1[s] + (long double)P.be[s] * w2[s];
            yp += br;
            if (n > 0) l1b += fabsl(br);
            const long double w0 = xin + (long double)P.na1[s] * w1[s] + (long double)P.na2[s] * w2[s];
            w2[s] = w1[s];
            w1[s] = w0;
        }
        hmax = std::max(hmax, fabsl(xc));
        emax = std::max(emax, fabsl(xc - yp));
        l1h += fabsl(xc);
    }
    if (!(hmax > 0.0L) || !std::isfinite((double)emax) || !std::isfinite((double)l1b)) return false;
    P.ir_err = (double)(emax / hmax);
    P.kappa = (double)(l1b / l1h);
    P.l1h = (double)l1h;
    return P.ir_err <= 1e-12 && P.kappa <= 1e3;
}

// V32 (see the kernel): may the from-rest end states of T-sample chunks of THIS filter be formed in float32?  The error such a state carries reaches the
// outputs of the next chunks through the output taps, amplified by whatever cancels between the branches -- no norm of the expansion predicts it (an
// elliptic band-pass with a cancellation factor of 2.8 shows 1.6e-6, one with 3.7 shows 4e-7), so it is MEASURED: the float32 chain of the matrix
// instruction (acc = fmaf(G[t], x[t], acc), oldest sample first, G rounded to float32 -- bit for bit what v_mfma_f32_16x16x4_f32 computes) against the
// exact from-rest state, on the inputs that are worst for it -- coherent ones: DC, the Nyquist alternation, a tone on every section's resonance -- and on
// noise; the state errors are carried from chunk to chunk by the exact transition and through the output taps sample by sample.  Returned: the worst
// output error over the probes, relative to the probe's output peak (or, for stop-band probes, 1 % of the forward bound -- the same floor the tests use).
static double par_v32_probe(const ParPlan &P, int T)
{
    const int N = P.nsec, NCH = 24, n = NCH * T;
    std::vector<float> g32((size_t)2 * N * T);
    std::vector<double> gd((size_t)2 * N * T);
    for (int k = 0; k < N; ++k) {
        long double g0 = 1.0L, g1 = 0.0L;
        for (int t = T - 1; t >= 0; --t) {
            gd[((size_t)2 * k) * T + t] = (double)g0;
            gd[((size_t)2 * k + 1) * T + t] = (double)g1;
            g32[((size_t)2 * k) * T + t] = (float)(double)g0;
            g32[((size_t)2 * k + 1) * T + t] = (float)(double)g1;
            const long double g2 = -P.a1[k] * g0 - P.a2[k] * g1;
            g1 = g0;
            g0 = g2;
        }
    }
    std::vector<std::vector<float>> probes;
    {
        std::vector<float> x((size_t)n);
        unsigned long long lcg = 0x2545F4914F6CDD1Dull;
        for (int i = 0; i < n; ++i) {
            double a = 0.0;
            for (int q = 0; q < 4; ++q) {
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                a += (double)(lcg >> 11) / 9007199254740992.0 - 0.5;
            }
            x[i] = (float)(a * 1.7320508075688772);
        }
        probes.push_back(x);
        for (int i = 0; i < n; ++i) x[i] = 1.0f;
        probes.push_back(x);
        for (int i = 0; i < n; ++i) x[i] = (i & 1) ? -1.0f : 1.0f;
        probes.push_back(x);
        for (int k = 0; k < N; ++k) {
            const double a1 = (double)P.a1[k], a2 = (double)P.a2[k];
            if (!(a2 > 0.0) || a1 * a1 >= 4.0 * a2) continue;   // (real poles: DC / Nyquist cover them)
            const double th = std::acos(std::max(-1.0, std::min(1.0, -a1 / (2.0 * std::sqrt(a2)))));
            for (int i = 0; i < n; ++i) x[i] = (float)std::cos(th * i);
            probes.push_back(x);
        }
    }
    double worst = 0.0;
    for (const auto &x : probes) {
        // the exact output (the parallel form in double, straight through) for the scale; the error by linearity: the state errors alone, carried exactly
        double ymax = 0.0, xmax = 0.0;
        {
            std::vector<double> w1((size_t)N, 0.0), w2((size_t)N, 0.0);
            for (int i = 0; i < n; ++i) {
                double yv = P.gamma * (double)x[i];
                for (int k = 0; k < N; ++k) {
                    yv += P.al[k] * w1[k] + P.be[k] * w2[k];
                    const double w0 = (double)x[i] + P.na1[k] * w1[k] + P.na2[k] * w2[k];
                    w2[k] = w1[k];
                    w1[k] = w0;
                }
                ymax = std::max(ymax, std::fabs(yv));
                xmax = std::max(xmax, (double)std::fabs(x[i]));
            }
        }
        std::vector<double> e1((size_t)N, 0.0), e2((size_t)N, 0.0);   // error of (w[n-1], w[n-2]) at the start of the current chunk
        double emax = 0.0;
        for (int j = 0; j < NCH; ++j) {
            // outputs of chunk j see the start-state error through the taps; the error state runs the homogeneous recurrence
            std::vector<double> f1 = e1, f2 = e2;
            for (int t = 0; t < T; ++t) {
                double ev = 0.0;
                for (int k = 0; k < N; ++k) {
                    ev += P.al[k] * f1[k] + P.be[k] * f2[k];
                    const double f0 = P.na1[k] * f1[k] + P.na2[k] * f2[k];
                    f2[k] = f1[k];
                    f1[k] = f0;
                }
                emax = std::max(emax, std::fabs(ev));
            }
            // this chunk's from-rest end state: the float32 chain against the double sum of the same products with the unrounded G
            for (int k = 0; k < N; ++k) {
                for (int c = 0; c < 2; ++c) {
                    const float *gf = g32.data() + ((size_t)2 * k + c) * T;
                    const double *ge = gd.data() + ((size_t)2 * k + c) * T;
                    float acc = 0.0f;
                    long double ex = 0.0L;
                    for (int t = 0; t < T; ++t) {
                        acc = std::fmaf(gf[t], x[(size_t)j * T + t], acc);
                        ex += (long double)ge[t] * (long double)x[(size_t)j * T + t];
                    }
                    (c == 0 ? f1[k] : f2[k]) += (double)acc - (double)ex;   // e_(j+1) = Phi e_j + delta_j (f holds Phi e_j now)
                }
            }
            e1 = f1;
            e2 = f2;
        }
        const double scale = std::max(ymax, 1e-2 * P.l1h * xmax);
        if (!(scale > 0.0) || !std::isfinite(emax)) return 1.0;
        worst = std::max(worst, emax / scale);
    }
    return worst;
}
constexpr double kParV32Limit = 5e-7;   // of the 1e-6 the float32 contract allows: the rest stays with the recurrence, the output rounding and the inputs no probe covers

struct M2 { long double m[4]; };
M2 m2mul(const M2 &x, const M2 &y)
{
    return M2{{x.m[0] * y.m[0] + x.m[1] * y.m[2], x.m[0] * y.m[1] + x.m[1] * y.m[3], x.m[2] * y.m[0] + x.m[3] * y.m[2],
               x.m[2] * y.m[1] + x.m[3] * y.m[3]}};
}
long double m2max(const M2 &x) { return std::max(std::max(fabsl(x.m[0]), fabsl(x.m[1])), std::max(fabsl(x.m[2]), fabsl(x.m[3]))); }

int par_tables(ParPlan &P, ParTables &tb, int T, int chunks, long double negl, int kmax, hipStream_t s)
{
    const int LV = chunks == 64 ? 6 : 5;   // scan levels inside a wave segment of `chunks` chunks
    const int N = P.nsec;
    tb.T = T;
    std::vector<double> lvl((size_t)6 * N * 4), psi((size_t)(kParMaxK - 1) * N * 4), gt((size_t)T * 16, 0.0);
    long double lvmax[7] = {0}, psimax[kParMaxK + 1] = {0};
    for (int k = 0; k < N; ++k) {
        // one-sample zero-input transition of (w[n-1], w[n-2]);  Phi = its T-th power
        M2 one{{-P.a1[k], -P.a2[k], 1.0L, 0.0L}}, Phi{{1.0L, 0.0L, 0.0L, 1.0L}}, sq = one;
        for (int e = T; e; e >>= 1) {
            if (e & 1) Phi = m2mul(Phi, sq);
            sq = m2mul(sq, sq);
        }
        M2 pw = Phi;
        for (int l = 0; l <= LV; ++l) {
            lvmax[l] = std::max(lvmax[l], m2max(pw));
            if (!std::isfinite((double)m2max(pw))) return 1;
            if (l < LV)
                for (int i = 0; i < 4; ++i) lvl[((size_t)l * N + k) * 4 + i] = (double)pw.m[i];
            if (l < LV) pw = m2mul(pw, pw);
        }
        const M2 Psi = pw;   // Phi^chunks: the transition over one wave segment
        M2 pk = Psi;
        for (int m = 1; m <= kParMaxK; ++m) {
            psimax[m] = std::max(psimax[m], m2max(pk));
            if (m < kParMaxK)
                for (int i = 0; i < 4; ++i) psi[((size_t)(m - 1) * N + k) * 4 + i] = (double)pk.m[i];
            pk = m2mul(pk, Psi);
        }
        // G rows 2k, 2k+1: g[T-1-t], g[T-2-t], g = impulse response of 1 / A_k; as the MFMA A operand of step t / 4:
        // lane l holds row l & 15, column 4 (t / 4) + (l >> 4)
        long double g0 = 1.0L, g1 = 0.0L;   // g[i], g[i-1]
        for (int t = T - 1; t >= 0; --t) {
            const size_t at = (size_t)(t / 4) * 64 + (size_t)(t % 4) * 16;
            gt[at + 2 * k] = (double)g0;
            gt[at + 2 * k + 1] = (double)g1;
            const long double g2 = -P.a1[k] * g0 - P.a2[k] * g1;
            g1 = g0;
            g0 = g2;
        }
    }
    tb.n_lv = LV;
    for (int l = LV; l >= 0; --l)
        if (lvmax[l] < negl) tb.n_lv = std::min(tb.n_lv, l);
    tb.K = 0;
    for (int m = 1; m <= kmax; ++m)
        if (psimax[m] < negl) { tb.K = m; break; }
    if (tb.K == 0) return 1;   // remembers more than kParMaxK segments
    if (tb.n_lv < LV) tb.K = 1;
    SK_HIP(hipMalloc((void **)&tb.gt_dev, gt.size() * 8));
    SK_HIP(hipMalloc((void **)&tb.lvl_dev, lvl.size() * 8));
    SK_HIP(hipMalloc((void **)&tb.psi_dev, psi.size() * 8));
    SK_HIP(hipMemcpyAsync(tb.gt_dev, gt.data(), gt.size() * 8, hipMemcpyHostToDevice, s));
    SK_HIP(hipMemcpyAsync(tb.lvl_dev, lvl.data(), lvl.size() * 8, hipMemcpyHostToDevice, s));
    SK_HIP(hipMemcpyAsync(tb.psi_dev, psi.data(), psi.size() * 8, hipMemcpyHostToDevice, s));
    SK_HIP(hipStreamSynchronize(s));
    return SKDSP_OK;
}
}  // namespace

// host-only: the expansion of a handle's cascade (tests; no GPU needed).  out = [c0, (a1, a2, r0, r1) x nsec, kappa, ir_err]
int iir_par_expand_host(const double *coef, int nsec, double *out, int *accepted)
{
    ParPlan P;
    P.nsec = nsec;
    const bool ok = nsec >= 1 && nsec <= 8 && par_expand(coef, nsec, P);
    if (accepted) *accepted = ok ? 1 : 0;
    if (out) {
        out[0] = (double)P.c0;
        for (int k = 0; k < nsec && k < 8; ++k) {
            out[1 + 4 * k] = (double)P.a1[k];
            out[2 + 4 * k] = (double)P.a2[k];
            out[3 + 4 * k] = (double)P.r0[k];
            out[4 + 4 * k] = (double)P.r1[k];
        }
        out[1 + 4 * nsec] = P.kappa;
        out[2 + 4 * nsec] = P.ir_err;
        // the float32 from-rest states of 128- and 96-sample chunks (V32): the worst probe error, admitted below kParV32Limit
        out[3 + 4 * nsec] = ok ? par_v32_probe(P, SK_PAR_T32) : 1.0;
        out[4 + 4 * nsec] = ok ? par_v32_probe(P, 96) : 1.0;
    }
    return SKDSP_OK;
}

// .dn: can a segment's kept outputs be gathered in the wave's stage image (see ParArgs::dec_compact)?
// (lean: the 96-sample kernels put a kept output into its slot straight from the sum -- no unit to pick it from, so M may be below the samples of a unit)
template <typename IO, bool CPLX> static bool par_dec_compact(int dec, int64_t seg_samples, bool lean = false, int64_t image_bytes = 0)
{
    constexpr int elems = 16 / (int)sizeof(IO), ls = CPLX ? 2 : 1;
    const int64_t slots = (seg_samples / dec + 2) * ls;
    const int64_t bytes = (slots + slots / 32 + 2) * (int64_t)sizeof(IO);
    return (lean || dec >= elems) && bytes <= (image_bytes ? image_bytes : (int64_t)64 * Stage<IO>::pitch * (int64_t)sizeof(IO)) && opt().iir_dn_compact;
}

// .dn with dec below the samples of a 16-byte unit (float32 / complex64, dec = 2, 3): gathered in two ranges of chunks behind the recurrence
template <typename IO, bool CPLX> static bool par_dec_rounds(int dec)
{
    return sizeof(IO) == 4 && (dec == 2 || dec == 3) && opt().iir_dn_compact;
}


// UPJ table of a plan for the factor L (see the kernel): rows c A^j, j = 0 .. L - 1, c = (al, be) of every section; then the 2 x 2 blocks of A^L
static int par_upj_table(ParPlan &P, int L, const double **out)
{
    for (auto &u : P.upj)
        if (u.up == L) { *out = u.dev; return SKDSP_OK; }
    const int N = P.nsec, D = 2 * N;
    // layout: rows 0 .. L - 1; up to 4 biquads (the kernels that read rows in pairs) row 0 once more; the blocks of A^L
    const size_t rows = (size_t)L + (N <= 4 ? 1 : 0);
    std::vector<double> tab(rows * D + (size_t)N * 4);
    for (int k = 0; k < N; ++k) {
        const long double A[4] = {-P.a1[k], -P.a2[k], 1.0L, 0.0L};   // (w[n-1], w[n-2]) -> (w[n], w[n-1]) without input
        long double c0 = (long double)P.al[k], c1 = (long double)P.be[k];   // the row c A^j
        long double M[4] = {1.0L, 0.0L, 0.0L, 1.0L};                  // A^j
        for (int j = 0; j < L; ++j) {
            tab[(size_t)j * D + 2 * k] = (double)c0;
            tab[(size_t)j * D + 2 * k + 1] = (double)c1;
            const long double n0 = c0 * A[0] + c1 * A[2], n1 = c0 * A[1] + c1 * A[3];
            c0 = n0; c1 = n1;
            const long double m0 = M[0] * A[0] + M[1] * A[2], m1 = M[0] * A[1] + M[1] * A[3], m2 = M[2] * A[0] + M[3] * A[2], m3 = M[2] * A[1] + M[3] * A[3];
            M[0] = m0; M[1] = m1; M[2] = m2; M[3] = m3;
        }
        if (N <= 4) {
            tab[(size_t)L * D + 2 * k] = tab[2 * k];
            tab[(size_t)L * D + 2 * k + 1] = tab[2 * k + 1];
        }
        for (int i = 0; i < 4; ++i) tab[rows * D + 4 * k + i] = (double)M[i];   // A^L
    }
    double *dev = nullptr;
    SK_HIP(hipMalloc((void **)&dev, tab.size() * 8));
    SK_HIP(hipMemcpy(dev, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    P.upj.push_back(ParPlan::UpJump{L, dev});
    *out = dev;
    return SKDSP_OK;
}

template <typename IO, bool CPLX, int TT = 0, bool UPJ = false, int UPS = 0>
static int launch_par_impl(IirHandle *h, ParPlan *p, ParTables &tb, const void *x, int64_t n, int nrow, int64_t x_stride, int64_t y_stride,
                           void *y, hipStream_t s, int dec, int up);

template <typename IO, bool CPLX, int TT = 0, bool UPJ = false, int UPS = 0>
static int launch_par(IirHandle *h, ParPlan *p, ParTables &tb, const void *x, int64_t n, int nrow, int64_t x_stride, int64_t y_stride,
                      void *y, hipStream_t s, int dec, int up = 1)
{
#ifdef SK_PAR_DEV_F32REAL   // (developer builds: float32 real signals only -- a third of the instantiations)
    if constexpr (sizeof(IO) == 8 || CPLX) return 1;
    else
#endif
    return launch_par_impl<IO, CPLX, TT, UPJ, UPS>(h, p, tb, x, n, nrow, x_stride, y_stride, y, s, dec, up);
}

template <typename IO, bool CPLX, int TT, bool UPJ, int UPS>
static int launch_par_impl(IirHandle *h, ParPlan *p, ParTables &tb, const void *x, int64_t n, int nrow, int64_t x_stride, int64_t y_stride,
                           void *y, hipStream_t s, int dec, int up)
{
    note_path("iir_par");   // (here, not in iir_par_launch: that function returns 1 -- nothing launched -- for every call the parallel form does not take)
    const int T = tb.T;
    // V32 (see the kernel): float32 / complex64 signals through 7 - 8 biquads, once the probe has admitted this filter at this chunk length
    bool v32 = false;
    // (not for .up by 8 or more through the general kernel: its from-rest states are formed per lane on the vector ALU from the FLOAT64 table -- `sparse` in the kernel)
    if (sizeof(IO) == 4 && !UPJ && h->nsec >= 7 && opt().iir_par_v32 > 0 && !(UPS == 0 && dec <= 1 && up >= 8)) {
        if (tb.v32 == 0) {
            tb.v32_err = par_v32_probe(*p, T);
            tb.v32 = tb.v32_err <= kParV32Limit ? 1 : -1;
        }
        v32 = tb.v32 == 1 || opt().iir_par_v32 >= 2;
        if (v32) note_path("iir_par_v32");
    }
    const int64_t S = (int64_t)(CPLX ? 32 : 64) * T;   // samples per wave segment
    const int64_t nseg = (n + S - 1) / S;
    SK_CHECK(nseg * nrow < (1 << 30), SKDSP_ERR_BADARG, "iir: too many segments");
    const int total = (int)(nseg * nrow);
    if (!p->ticket_dev) {
        SK_HIP(hipMalloc((void **)&p->ticket_dev, 8 * kParTickets));
        SK_HIP(hipMemsetAsync(p->ticket_dev, 0, 8 * kParTickets, s));
        p->ticket_count = 0;
    }
    const size_t need = (size_t)total * (CPLX ? 64 : 32) * 8;
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
    ParArgs a;
    a.x = x; a.y = y; a.n = n; a.x_stride = x_stride; a.y_stride = y_stride;
    a.nseg = (int)nseg; a.total = total;
    a.lb = p->lbg_dev; a.ticket = p->ticket_dev; a.ticket_base = p->ticket_count;
    a.epoch = ++p->epoch;
    if (a.epoch == 0) a.epoch = ++p->epoch;
    a.n_lv = tb.n_lv; a.K = tb.K;
    a.err = async_err_dev(kAsyncErrIirLookback);
    SK_CHECK(a.err, SKDSP_ERR_HIP, "iir: no host-mapped error word");
    a.aligned = ((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (nrow == 1 || ((x_stride * sizeof(IO)) % 16 == 0 && (y_stride * sizeof(IO)) % 16 == 0))) ? 1 : 0;
    a.dec = dec > 1 ? dec : 1;
    a.dec_magic = a.dec > 1 ? (unsigned)((((unsigned long long)1 << 32) + a.dec - 1) / a.dec) : 0u;
    // (M = 2 on 96-sample chunks, more than 4 biquads, float32 / complex64: the DECM = 3 kernels with their larger image)
    const bool big_m2 = TT == 96 && sizeof(IO) == 4 && !UPJ && UPS == 0 && a.dec == 2 && h->nsec > 4 && opt().iir_dn_t96 != 3;
    a.dec_compact = a.dec > 1 && par_dec_compact<IO, CPLX>(a.dec, S, TT == 96, big_m2 ? kParStageM2 : 0) ? 1 : 0;
    a.dec_rounds = !a.dec_compact && par_dec_rounds<IO, CPLX>(a.dec) ? 2 : 1;
    a.up = up > 1 ? up : 1;
    a.up_magic = a.up > 1 ? (unsigned)((((unsigned long long)1 << 32) + a.up - 1) / a.up) : 0u;
    a.n_in = a.up > 1 ? n / a.up : n;
    a.n_keep = (n / a.dec) * a.dec;
    const double *upj_tab = nullptr;
    if constexpr (UPJ) {
        const int rc = par_upj_table(*p, a.up, &upj_tab);
        if (rc) return rc;
    }
    {
        const int64_t step = (int64_t)(64 / Stage<IO>::segs) * T;   // samples between a lane's staged segments
        a.dec_dq = (int)(step / a.dec);
        a.dec_dr = (int)(step % a.dec);
    }
    // (whole rounds of kParTickets workgroups, so that every dispenser advances by the same amount: the surplus workgroups
    // draw segments beyond the last and leave)
    const unsigned grid = (unsigned)(((total + 3) / 4 + kParTickets - 1) / kParTickets * kParTickets);
    p->ticket_count += (unsigned long long)(grid / kParTickets);
#define SK_PAR(N)                                                                                                       \
    case N: {                                                                                                           \
        ParCoef<N> cf;                                                                                                  \
        for (int k = 0; k < N; ++k) { cf.na1[k] = p->na1[k]; cf.na2[k] = p->na2[k]; cf.al[k] = p->al[k]; cf.be[k] = p->be[k]; } \
        cf.gamma = p->gamma;                                                                                            \
        if (a.dec > 1 && a.dec_rounds > 1) {                                                                            \
            if constexpr (sizeof(IO) == 4 && !UPJ && UPS == 0)   /* (the .up launchers never decimate: no decimating kernels on their account) */ \
                hipLaunchKernelGGL((iir_par_kernel<N, IO, 2, CPLX, TT>), dim3(grid), dim3(kIirThreads), 0, s, a, cf,    \
                                   (const double *)tb.gt_dev, (const double *)tb.lvl_dev, (const double *)tb.psi_dev, 0ull, (const double *)nullptr);  \
        } else if (a.dec > 1 && big_m2 && a.dec_compact) {                                                              \
            if constexpr (TT == 96 && sizeof(IO) == 4 && N > 4 && !UPJ && UPS == 0)                                     \
                hipLaunchKernelGGL((iir_par_kernel<N, IO, 3, CPLX, TT>), dim3(grid), dim3(kIirThreads), 0, s, a, cf,    \
                                   (const double *)tb.gt_dev, (const double *)tb.lvl_dev, (const double *)tb.psi_dev, 0ull, (const double *)nullptr);  \
        } else if (a.dec > 1) {                                                                                         \
            if constexpr (!UPJ && UPS == 0) {                                                                           \
                if constexpr (N >= 7 && sizeof(IO) == 4) {                                                              \
                    if (v32) {                                                                                          \
                        hipLaunchKernelGGL((iir_par_kernel<N, IO, 1, CPLX, TT, false, 0, true>), dim3(grid), dim3(kIirThreads), 0, s, a, cf, \
                                           (const double *)tb.gt_dev, (const double *)tb.lvl_dev, (const double *)tb.psi_dev, 0ull, (const double *)nullptr); \
                        break;                                                                                          \
                    }                                                                                                   \
                }                                                                                                       \
                hipLaunchKernelGGL((iir_par_kernel<N, IO, 1, CPLX, TT>), dim3(grid), dim3(kIirThreads), 0, s, a, cf,    \
                                   (const double *)tb.gt_dev, (const double *)tb.lvl_dev, (const double *)tb.psi_dev, 0ull, (const double *)nullptr);  \
            }                                                                                                           \
        } else if constexpr (UPJ)                                                                                         \
            hipLaunchKernelGGL((iir_par_kernel<N, IO, 0, CPLX, TT, true>), dim3(grid), dim3(kIirThreads), 0, s, a, cf,  \
                               (const double *)tb.gt_dev, (const double *)tb.lvl_dev, (const double *)tb.psi_dev, 0ull, upj_tab); \
        else if constexpr (UPS != 0) {                                                                                  \
            if constexpr (N >= 7 && sizeof(IO) == 4) {                                                                  \
                if (v32) {                                                                                              \
                    hipLaunchKernelGGL((iir_par_kernel<N, IO, 0, CPLX, TT, false, UPS, true>), dim3(grid), dim3(kIirThreads), 0, s, a, cf, \
                                       (const double *)tb.gt_dev, (const double *)tb.lvl_dev, (const double *)tb.psi_dev, 0ull, (const double *)nullptr); \
                    break;                                                                                              \
                }                                                                                                       \
            }                                                                                                           \
            hipLaunchKernelGGL((iir_par_kernel<N, IO, 0, CPLX, TT, false, UPS>), dim3(grid), dim3(kIirThreads), 0, s, a, cf, \
                               (const double *)tb.gt_dev, (const double *)tb.lvl_dev, (const double *)tb.psi_dev, 0ull, (const double *)nullptr); \
        } else if constexpr (TT == 0) {                                                                                 \
            if constexpr (N >= 7 && sizeof(IO) == 4) {                                                                  \
                if (v32) {                                                                                              \
                    hipLaunchKernelGGL((iir_par_kernel<N, IO, 0, CPLX, 0, false, 0, true>), dim3(grid), dim3(kIirThreads), 0, s, a, cf, \
                                       (const double *)tb.gt_dev, (const double *)tb.lvl_dev, (const double *)tb.psi_dev, 0ull, (const double *)nullptr); \
                    break;                                                                                              \
                }                                                                                                       \
            }                                                                                                           \
            hipLaunchKernelGGL((iir_par_kernel<N, IO, 0, CPLX>), dim3(grid), dim3(kIirThreads), 0, s, a, cf,            \
                               (const double *)tb.gt_dev, (const double *)tb.lvl_dev, (const double *)tb.psi_dev, 0ull, (const double *)nullptr);      \
        }                                                                                                               \
        break;                                                                                                          \
    }
    switch (h->nsec) {
#ifdef SK_PAR_DEV_NSEC   // (developer builds: one cascade length only -- this file is the long pole of the build)
        SK_PAR(SK_PAR_DEV_NSEC)
#else
        SK_PAR(1) SK_PAR(2) SK_PAR(3) SK_PAR(4) SK_PAR(5) SK_PAR(6) SK_PAR(7)
        SK_PAR(8)
#endif
        default: SK_CHECK(false, SKDSP_ERR_UNSUPPORTED, "iir: parallel-form scan takes 1..8 biquads");
    }
#undef SK_PAR
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

// returns 1 when the parallel form does not apply to this handle / call (nothing was launched)
int iir_par_launch(IirHandle *h, const void *x, int64_t n, int nrow, int64_t x_stride, int64_t y_stride, void *y, hipStream_t s, int dec,
                   int interleaved, int up)
{
    if (interleaved && nrow != 1) return 1;
    // .up: x holds n / up samples; one row, no decimation; the exact-division trick of the staging covers up <= 4096
    if (up > 1 && (dec > 1 || nrow != 1 || up > 4096 || n % up != 0)) return 1;
    if (h->order != 2 || h->nsec < 1 || h->nsec > 8) return 1;
    if (!h->par) {
        h->par = new ParPlan();
        h->par->nsec = h->nsec;
        h->par->state = par_expand(h->coef.data(), h->nsec, *h->par) ? 1 : -1;
    }
    ParPlan *p = h->par;
    if (p->state != 1) return 1;
    const bool dbl = dtype_double(h->dtype);
    // .dn of float32 / complex64 signals by a divisor of 96: chunks of 96 samples, so that all lanes of a wave walk the same phase (see the kernel).
    // From M = 4 on the 96-sample kernel is the compact store in its lean form (DNL in the kernel: which samples are kept is wave-uniform).  Measured, 2^26 inputs
    // (_var/dn_t96.py, profiles/r05/iir_dn_lean.txt): order-8 Butterworth M = 4 .. 96 float32 0.105 - 0.122 -> 0.086 - 0.099 ms, complex64 0.198 - 0.223 -> 0.172 - 0.196;
    // 8 biquads: float32 - 3 .. - 6 %, complex64 - 3 % where 3 divides M (128-sample chunks then start on three phases) and + 4 .. + 7 % elsewhere.
    // Before the lean form: M = 2, 3, 6 only (M = 3 0.193 -> 0.173 ms, M = 4 + 7 .. 9 %).  Option iir_dn_t96 = 2: every divisor of 96; 0: never.
    const bool t96_pays = dec == 2 || dec == 3 || dec == 6 || (96 % dec == 0 && (h->nsec <= 4 || !interleaved || dec % 3 == 0));
    bool t96 = !dbl && dec > 1 && (((opt().iir_dn_t96 == 1 || opt().iir_dn_t96 == 3) && t96_pays) || (opt().iir_dn_t96 == 2 && 96 % dec == 0));
    // (the 96-sample kernels have no store but the gathering ones: the lean compact store where a segment's kept outputs fit the image -- from M = 3 on --, ranges of chunks for M = 2)
    if (t96 && !(interleaved ? par_dec_compact<float, true>(dec, (int64_t)32 * 96, true) || par_dec_rounds<float, true>(dec)
                             : par_dec_compact<float, false>(dec, (int64_t)64 * 96, true) || par_dec_rounds<float, false>(dec)))
        t96 = false;
    // .up by a divisor of 96 from 8 on (the reference default 12): the lean kernels whose state jumps from input sample to input sample (UPJ, chunks of 96 so
    // that every chunk starts on one).  Measured, same box (profiles/r05/iir_up_lean.txt): rate_change(12).up float32 0.101 -> 0.072 ms per 2^26 outputs; 8-biquad
    // elliptic by 12 0.113 -> 0.092 per 5e7 (complex64 0.216 -> 0.183); 5 biquads by 8 0.132 -> 0.096 (0.231 -> 0.164).  Option iir_up_jump = 0: never
    const bool upj = dec <= 1 && up >= 8 && 96 % up == 0 && opt().iir_up_jump >= 1;
    // .up by 3 (a stage of sigsys.interp24): the lean staging at the input rate (UPS in the kernel), on chunks of 96
    const bool ups3 = !dbl && dec <= 1 && up == 3 && opt().iir_up_lean;   // (float64: four images of 96 doubles per row and the table leave room for ONE workgroup per CU)
    t96 = t96 || upj || ups3;
    if (t96) {
        ParTables &t9 = p->tab[(dbl ? 6 : 4) + (interleaved ? 1 : 0)];
        if (t9.T == 0) {
            const int rc = par_tables(*p, t9, 96, interleaved ? 32 : 64, dbl ? 1e-30L : 1e-18L, par_max_k(dbl), s);
            if (rc < 0) return rc;
        }
        if (t9.K == 0) t96 = false;   // (the filter remembers more segments of this length than the look-back serves: the 128-sample chunks, if they do)
    }
    if (t96 && upj) {
        if (dbl)
            return interleaved ? launch_par<double, true, 96, true>(h, p, p->tab[7], x, n, 1, 0, 0, y, s, dec, up)
                               : launch_par<double, false, 96, true>(h, p, p->tab[6], x, n, nrow, x_stride, y_stride, y, s, dec, up);
        return interleaved ? launch_par<float, true, 96, true>(h, p, p->tab[5], x, n, 1, 0, 0, y, s, dec, up)
                           : launch_par<float, false, 96, true>(h, p, p->tab[4], x, n, nrow, x_stride, y_stride, y, s, dec, up);
    }
    if (t96 && ups3) {
        ParTables &t3 = p->tab[4 + (interleaved ? 1 : 0)];
        return interleaved ? launch_par<float, true, 96, false, 3>(h, p, t3, x, n, 1, 0, 0, y, s, dec, up)
                           : launch_par<float, false, 96, false, 3>(h, p, t3, x, n, nrow, x_stride, y_stride, y, s, dec, up);
    }
    if (upj || ups3) t96 = false;
    ParTables &tb = t96 ? p->tab[4 + (interleaved ? 1 : 0)] : p->tab[(dbl ? 1 : 0) + (interleaved ? 2 : 0)];
    if (tb.T == 0) {
        // negligibility as in iir_scan.hip: 1e-30 for float64 signals, 1e-18 for float32 signals (a tenth of an ulp of the
        // float64 state the dropped term would be added to)
        const int rc = par_tables(*p, tb, dbl ? SK_PAR_T32 / 2 : SK_PAR_T32, interleaved ? 32 : 64, dbl ? 1e-30L : 1e-18L, par_max_k(dbl), s);
        if (rc < 0) return rc;
    }
    if (tb.K == 0) return 1;
    // (interleaved signals have no decimating store here but the compact one)
    if (interleaved && dec > 1 && !(dbl ? par_dec_compact<double, true>(dec, (int64_t)32 * tb.T)
                                        : (par_dec_compact<float, true>(dec, (int64_t)32 * tb.T) || par_dec_rounds<float, true>(dec)))) return 1;
    if (t96)
        return interleaved ? launch_par<float, true, 96>(h, p, tb, x, n, 1, 0, 0, y, s, dec, up)
                           : launch_par<float, false, 96>(h, p, tb, x, n, nrow, x_stride, y_stride, y, s, dec, up);
    // .up by 2 / 4 (where the state jump does not pay; with 3 above, the stages of sigsys.interp24): staged at the input rate, the stuffed zeros known to the
    // compiler (UPS in the kernel).  Measured (profiles/r05/iir_up_lean.txt); option iir_up_lean = 0: the zero-stuffed image as for every other factor
    if (up == 2 && dec <= 1 && opt().iir_up_lean) {
        if (interleaved)
            return dbl ? launch_par<double, true, 0, false, 2>(h, p, tb, x, n, 1, 0, 0, y, s, dec, up)
                       : launch_par<float, true, 0, false, 2>(h, p, tb, x, n, 1, 0, 0, y, s, dec, up);
        return dbl ? launch_par<double, false, 0, false, 2>(h, p, tb, x, n, nrow, x_stride, y_stride, y, s, dec, up)
                   : launch_par<float, false, 0, false, 2>(h, p, tb, x, n, nrow, x_stride, y_stride, y, s, dec, up);
    }
    if (up == 4 && dec <= 1 && !dbl && opt().iir_up_lean)   // (a float64 chunk of 64 holds 16 inputs: half a staging piece)
        return interleaved ? launch_par<float, true, 0, false, 4>(h, p, tb, x, n, 1, 0, 0, y, s, dec, up)
                           : launch_par<float, false, 0, false, 4>(h, p, tb, x, n, nrow, x_stride, y_stride, y, s, dec, up);
    if (interleaved)
        return dbl ? launch_par<double, true>(h, p, tb, x, n, 1, 0, 0, y, s, dec, up) : launch_par<float, true>(h, p, tb, x, n, 1, 0, 0, y, s, dec, up);
    return dbl ? launch_par<double, false>(h, p, tb, x, n, nrow, x_stride, y_stride, y, s, dec, up)
               : launch_par<float, false>(h, p, tb, x, n, nrow, x_stride, y_stride, y, s, dec, up);
}

}  // namespace skdsp
