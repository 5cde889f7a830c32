// fir_mm.hip -- direct / polyphase FIR of float32 or complex64 signals with real taps as a Toeplitz
// matrix product on the FP32 matrix pipe (gfx950: v_mfma_f32_16x16x4_f32, exact f32 FMA chains at the
// FP32 vector peak -- which a v_pk_fma_f32 sliding-window kernel reaches to ~40 % at best).
//
// Serves the same reference calls as fir_direct.hip (multirate_helper.py:104-127 and
// downsample(up(x,L),M)):   y[m] = L * sum_t b[phi_c + L t] * x[i_c + q s - t],
//     m = c + L' s,  c = m mod L' (class),  L' = L/gcd, q = M/gcd,  phi_c = (c M) mod L, i_c = (c M) div L.
//
// 16 output ROWS are formed from DS = floor(16 / L') consecutive slots of all L' classes,
//     row r = L' ds + c  (ds < DS),   column N = slot block,   slot s = DS N + ds,
// so that the output index is  m = RS N + r  (RS = L' DS rows in use: a column block IS a contiguous run of
// the output) and the input index is  q DS N + U0 - u  with a lag u = t + U0 - i_c - q ds >= 0 that no
// longer depends on N.  Hence
//     Y[16 x N] = A[16 x K] * W[K x N],   A[r][u] = L b[phi_c + L (u - U0 + i_c + q ds)]  (0 outside the taps),
//                                          W[u][N] = x[q DS N + U0 - u],     K = T + U0,  T = ceil(P / L)
// A depends on the filter only: each lane keeps its K/4 A-operands (row l & 15, lag 4 ks + (l >> 4)) in
// registers for the whole launch.  W is read straight out of the LDS window (lane l: column l & 15, lag
// l >> 4 of the step), complex samples as one 8-byte read feeding two MFMAs (re, im).  The accumulator
// layout (col = lane & 15, row = 4 (lane >> 4) + reg) puts 4 consecutive outputs in every lane and a
// wave's 16 x 16 tile in 256 consecutive outputs: stores are full 2 KiB rows with no transposition.
// The window is stored with one pad element after every 8 (physical = e + (e >> 3)): the 16 columns of a
// B read are q DS elements apart, which without the skew lands them on a few banks for most strides
// (12 complex elements: 2-way, 16: 8-way; with it at most 2-way for every stride, none for 12).
#include "skdsp_internal.hpp"
#include <numeric>
#include <vector>

namespace skdsp {

typedef float v4f_mm __attribute__((ext_vector_type(4)));

struct MmArgs {
    int64_t n, n_hist, n_out;
    int q_ds;   // q * DS: input samples per column block
    int RS;     // rows in use = L' * DS (<= 16): outputs per column block
    int K4;     // K / 4 MFMA steps (K padded to a multiple of 4)
    int U0;     // lag offset (see above)
    int NS;     // column blocks per workgroup (multiple of 64)
    int win;    // staged samples per workgroup = q_ds * (NS - 1) + 4 * K4
    int L, M;   // of the call, for the re-evaluation of non-finite results (careful.hpp)
    CarefulFir cf;
};

__device__ __forceinline__ int mm_phys(int e) { return e + (e >> 3); }

typedef double v4d_mm __attribute__((ext_vector_type(4)));

// signal type -> scalar type, components, accumulator vector, MFMA
template <typename X> struct MmIo;
template <> struct MmIo<float> { using S = float; using V = v4f_mm; static constexpr int C = 1; };
template <> struct MmIo<float2> { using S = float; using V = v4f_mm; static constexpr int C = 2; };
template <> struct MmIo<double> { using S = double; using V = v4d_mm; static constexpr int C = 1; };
template <> struct MmIo<double2> { using S = double; using V = v4d_mm; static constexpr int C = 2; };

__device__ __forceinline__ v4f_mm mm_mfma(float a, float b, v4f_mm c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4d_mm mm_mfma(double a, double b, v4d_mm c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
template <typename S> __device__ __forceinline__ S mm_re(S v) { return v; }
__device__ __forceinline__ float mm_re(float2 v) { return v.x; }
__device__ __forceinline__ double mm_re(double2 v) { return v.x; }
__device__ __forceinline__ float mm_im(float2 v) { return v.y; }
__device__ __forceinline__ double mm_im(double2 v) { return v.y; }
template <typename X> __device__ __forceinline__ X mm_zero();
template <> __device__ __forceinline__ float mm_zero<float>() { return 0.f; }
template <> __device__ __forceinline__ double mm_zero<double>() { return 0.0; }
template <> __device__ __forceinline__ float2 mm_zero<float2>() { return make_float2(0.f, 0.f); }
template <> __device__ __forceinline__ double2 mm_zero<double2>() { return make_double2(0.0, 0.0); }
template <typename S> __device__ __forceinline__ void mm_put(S *y, S re, S) { *y = re; }
__device__ __forceinline__ void mm_put(float2 *y, float re, float im) { *y = make_float2(re, im); }
__device__ __forceinline__ void mm_put(double2 *y, double re, double im) { *y = make_double2(re, im); }

// The f64 MFMA returns rows (lane >> 4) + 4 reg instead of 4 (lane >> 4) + reg; the host permutes the
// rows of A for float64 / complex128 so that, either way, register i of lane group g is output row 4 g + i.
template <typename X, int K4B>
__global__ __launch_bounds__(256) void fir_mm_kernel(const X *__restrict__ x, const typename MmIo<X>::S *__restrict__ At, MmArgs a,
                                                     X *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int mm_nf;   // some result of this workgroup came out non-finite (see the end of the kernel)
    if (threadIdx.x == 0) mm_nf = 0;
    X *win = reinterpret_cast<X *>(smem_raw);
    using S = typename MmIo<X>::S;
    using V = typename MmIo<X>::V;
    constexpr bool CPLX = MmIo<X>::C == 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = 4 * a.K4;
    const int64_t S0 = (int64_t)blockIdx.x * a.NS;                 // first column block of this workgroup
    const int64_t g0 = (int64_t)a.q_ds * S0 + a.U0 - (K - 1);       // input index of win[0]

    // A operands of this lane for every step (zero beyond K4)
    S areg[K4B];  // a.K4 == K4B: the host pads the lag range with zero taps up to the instantiated size
#pragma unroll
    for (int ks = 0; ks < K4B; ++ks) areg[ks] = At[ks * 64 + lane];

    // ---- stage the window (zero outside [-n_hist, n)); 16-byte loads for interior workgroups ----
    {
        constexpr int VEC = 16 / (int)sizeof(X);
        const X *src = x + g0;
        const bool interior = g0 >= -a.n_hist && g0 + a.win <= a.n && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        if (interior) {
            const int nv = a.win / VEC;
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            for (int k0 = tid; k0 < nv; k0 += 256 * 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + 256 * u < nv) v[u] = s4[k0 + 256 * u];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + 256 * u < nv) {
                        const int i = (k0 + 256 * u) * VEC;  // VEC <= 4 elements starting at a multiple of VEC share a pad group
                        X *dst = win + mm_phys(i);
                        const X *e = reinterpret_cast<const X *>(&v[u]);
#pragma unroll
                        for (int t = 0; t < VEC; ++t) dst[t] = e[t];
                    }
            }
            for (int i = nv * VEC + tid; i < a.win; i += 256) win[mm_phys(i)] = src[i];
        } else {
            for (int i0 = tid; i0 < a.win; i0 += 256 * 8) {
                X v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 256 * u;
                    const int64_t g = g0 + i;
                    v[u] = mm_zero<X>();
                    if (i < a.win && g >= -a.n_hist && g < a.n) v[u] = x[g];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 256 * u < a.win) win[mm_phys(i0 + 256 * u)] = v[u];
            }
        }
    }
    __syncthreads();

    const int ncol = lane & 15, klane = lane >> 4;
    const int ntiles = a.NS / 16;
    for (int tile = wave; tile < ntiles; tile += 4) {
        // W[u][N]: window index of (column N, lag u) = q_ds * Nloc + (K - 1) - u
        const int eb = a.q_ds * (tile * 16 + ncol) + (K - 1) - klane;
        // NA interleaved accumulator sets: independent MFMA chains, and partial sums of K / NA terms each, added as a tree
        // (a single f32 chain over all lags sits at 5e-7 of the float64 result for ~150 lags on NOISE; on coherent inputs
        // -- DC through all-positive taps -- every addition of a chain rounds the same way and the error grows with the
        // chain length: 8 sets instead of 4 for the float32 engines halve it; the float64 engine keeps 4)
        constexpr int NA = sizeof(S) == 4 ? 8 : 4;
        V ar[NA], ai[NA];
#pragma unroll
        for (int c = 0; c < NA; ++c) ar[c] = ai[c] = V{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < K4B; ++ks) {
            const X b = win[mm_phys(eb - 4 * ks)];
            ar[ks % NA] = mm_mfma(areg[ks], mm_re(b), ar[ks % NA]);
            if constexpr (CPLX) ai[ks % NA] = mm_mfma(areg[ks], mm_im(b), ai[ks % NA]);
        }
#pragma unroll
        for (int w = NA / 2; w >= 1; w /= 2)
#pragma unroll
            for (int c = 0; c < w; ++c) {
                ar[c] = ar[c] + ar[c + w];
                ai[c] = ai[c] + ai[c + w];
            }
        const V ar0 = ar[0];
        const V ai0 = ai[0];
        // rows 4 (lane >> 4) + i of column N: outputs m = RS N + row
        const int64_t N = S0 + tile * 16 + ncol;
        const int row0 = 4 * klane;
        const int64_t m0 = (int64_t)a.RS * N + row0;
        {   // a non-finite result: did a real tap meet the sample, or only the zero padding of the lag range?  Looked at behind the loop (careful.hpp)
            bool bad = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) bad |= not_finite((S)ar0[i]) || (CPLX && not_finite((S)ai0[i]));
            if (__builtin_expect(__any(bad), 0)) mm_nf = 1;
        }
        if (a.RS == 16 && m0 + 4 <= a.n_out && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
            // 4 consecutive outputs per lane: 16-byte stores
            S buf[4 * MmIo<X>::C];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                buf[MmIo<X>::C * i] = ar0[i];
                if constexpr (CPLX) buf[2 * i + 1] = ai0[i];
            }
            constexpr int NV = 4 * (int)sizeof(X) / 16;
            float4 *dst = reinterpret_cast<float4 *>(y + m0);
#pragma unroll
            for (int v = 0; v < NV; ++v) dst[v] = reinterpret_cast<const float4 *>(buf)[v];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (row0 + i < a.RS && m0 + i < a.n_out) mm_put(y + m0 + i, (S)ar0[i], (S)ai0[i]);
        }
    }
    // non-finite results of this workgroup's outputs m in [RS S0, RS (S0 + NS)): re-evaluated by the reference's sum (careful.hpp)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (__builtin_expect(*reinterpret_cast<volatile int *>(&mm_nf) != 0, 0))
        careful_fir_recheck<S, CPLX>(x, y, a.n_hist, a.n_out, (int64_t)a.RS * S0, (int64_t)a.RS * a.NS, a.L, a.M, a.cf, tid);
}

// A-operand table of one (L, M): At[ks][lane] = A[row = lane & 15][u = 4 ks + (lane >> 4)]
static int get_mm_table(FirHandle *h, int L, int M, const FirHandle::MmTab **out)
{
    for (auto &t : h->mm)
        if (t.L == L && t.M == M) { *out = &t; return SKDSP_OK; }
    const int g = std::gcd(L, M), Lp = L / g, q = M / g;
    const int P = h->ntaps, T = (P + L - 1) / L;
    const int DS = 16 / Lp, RS = Lp * DS;
    int imax = 0;
    for (int c = 0; c < Lp; ++c) imax = std::max(imax, (int)(((int64_t)c * M) / L));
    const int U0 = imax + q * (DS - 1);
    const int K4 = ((T + U0 + 3) / 4 + 3) / 4 * 4;  // padded to the kernel instantiations (multiples of 4 steps)
    const bool dbl = dtype_double(h->dtype);
    std::vector<double> host((size_t)K4 * 64, 0.0);
    for (int r = 0; r < RS; ++r) {
        const int ds = r / Lp, c = r % Lp;
        const int64_t cm = (int64_t)c * M;
        const int phi = (int)(cm % L), ic = (int)(cm / L);
        for (int u = 0; u < 4 * K4; ++u) {
            const int t = u - U0 + ic + q * ds;
            if (t < 0 || t >= T) continue;
            const int k = phi + L * t;
            if (k >= P) continue;
            // MFMA row that must hold output row r: the f64 instruction returns rows (lane >> 4) + 4 reg
            const int row = dbl ? (r % 4) * 4 + r / 4 : r;
            host[(size_t)(u / 4) * 64 + (size_t)(u % 4) * 16 + row] = (double)L * h->taps_host[k];
        }
    }
    std::vector<float> hostf;
    if (!dbl) hostf.assign(host.begin(), host.end());
    FirHandle::MmTab t;
    t.L = L; t.M = M; t.Lp = Lp; t.q = q; t.DS = DS; t.RS = RS; t.U0 = U0; t.K4 = K4; t.At = nullptr;
    const size_t tbytes = host.size() * (dbl ? 8 : 4);
    SK_HIP(hipMalloc(&t.At, tbytes));
    SK_HIP(hipMemcpy(t.At, dbl ? (const void *)host.data() : (const void *)hostf.data(), tbytes, hipMemcpyHostToDevice));
    h->mm.push_back(t);
    *out = &h->mm.back();
    return SKDSP_OK;
}

bool fir_mm_supported(const FirHandle *h, int L, int M, int64_t n_out)
{
    if (h->taps_complex || h->dtype == SKDSP_C128) return false;  // complex128: the sliding-window kernel measured faster
    const int kmax = dtype_double(h->dtype) ? 48 : 96;  // A operands in registers: 1 (float) or 2 (double) VGPRs per step
    const int g = std::gcd(L, M), Lp = L / g, q = M / g;
    if (Lp > 16) return false;
    const int T = (h->ntaps + L - 1) / L, DS = 16 / Lp;
    const int64_t imax = ((int64_t)(Lp - 1) * M) / L;
    const int64_t K = T + imax + (int64_t)q * (DS - 1);
    if (K > 4 * kmax - 12) return false;                  // A operands must fit the register file
    const int64_t win = (int64_t)q * DS * 63 + K + 32;    // smallest workgroup tile (NS = 64)
    if (win * 9 / 8 * (int64_t)dtype_size(h->dtype) > 63 * 1024) return false;
    return n_out >= 16 * 64;
}

int fir_mm_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, int L, int M, int64_t n_out, void *y, hipStream_t s)
{
    note_path("fir_mm");
    if (n_out <= 0) return SKDSP_OK;
    const FirHandle::MmTab *t = nullptr;
    int rc = get_mm_table(h, L, M, &t);
    if (rc) return rc;
    const size_t esz = dtype_size(h->dtype);
    MmArgs a;
    a.n = n; a.n_hist = n_hist; a.n_out = n_out;
    a.q_ds = t->q * t->DS; a.RS = t->RS; a.K4 = t->K4; a.U0 = t->U0;
    // column blocks per workgroup: as many as a 64 KiB window holds (2 workgroups per CU), at most 512
    int NS = 256;
    while (NS > 64 && ((size_t)a.q_ds * (NS - 1) + 4 * a.K4) * 9 / 8 * esz > (size_t)63 * 1024) NS -= 64;
    const int64_t ncols = (n_out + a.RS - 1) / a.RS;
    while (NS > 64 && (ncols + NS - 1) / NS < 2 * ctx().num_cus) NS -= 64;  // small problems: more workgroups
    a.NS = NS;
    a.win = a.q_ds * (NS - 1) + 4 * a.K4;
    a.L = L; a.M = M;
    if ((rc = fir_careful(h, &a.cf))) return rc;
    const size_t lds = ((((size_t)a.win + (size_t)a.win / 8 + 2) * esz + 15) & ~(size_t)15) + 64;
    const unsigned grid = (unsigned)((ncols + NS - 1) / NS);
#define SK_MM(XT, KB)                                                                                              \
    hipLaunchKernelGGL((fir_mm_kernel<XT, KB>), dim3(grid), dim3(256), lds, s, (const XT *)x, (const typename MmIo<XT>::S *)t->At, a, (XT *)y)
#define SK_MMK(XT)                                                       \
    switch (a.K4) {                                                      \
    case 4: SK_MM(XT, 4); break;                                      \
    case 8: SK_MM(XT, 8); break;                                      \
    case 12: SK_MM(XT, 12); break;                                      \
    case 16: SK_MM(XT, 16); break;                                      \
    case 20: SK_MM(XT, 20); break;                                      \
    case 24: SK_MM(XT, 24); break;                                      \
    case 28: SK_MM(XT, 28); break;                                      \
    case 32: SK_MM(XT, 32); break;                                      \
    case 36: SK_MM(XT, 36); break;                                      \
    case 40: SK_MM(XT, 40); break;                                      \
    case 44: SK_MM(XT, 44); break;                                      \
    case 48: SK_MM(XT, 48); break;                                      \
    case 52: SK_MM(XT, 52); break;                                      \
    case 56: SK_MM(XT, 56); break;                                      \
    case 60: SK_MM(XT, 60); break;                                      \
    case 64: SK_MM(XT, 64); break;                                      \
    case 68: SK_MM(XT, 68); break;                                      \
    case 72: SK_MM(XT, 72); break;                                      \
    case 76: SK_MM(XT, 76); break;                                      \
    case 80: SK_MM(XT, 80); break;                                      \
    case 84: SK_MM(XT, 84); break;                                      \
    case 88: SK_MM(XT, 88); break;                                      \
    case 92: SK_MM(XT, 92); break;                                      \
    case 96: SK_MM(XT, 96); break;                                      \
    default: SK_CHECK(false, SKDSP_ERR_UNSUPPORTED, "fir_mm: %d steps", a.K4); \
    }
#define SK_MMKD(XT)                                                      \
    switch (a.K4) {                                                      \
    case 4: SK_MM(XT, 4); break;                                      \
    case 8: SK_MM(XT, 8); break;                                      \
    case 12: SK_MM(XT, 12); break;                                      \
    case 16: SK_MM(XT, 16); break;                                      \
    case 20: SK_MM(XT, 20); break;                                      \
    case 24: SK_MM(XT, 24); break;                                      \
    case 28: SK_MM(XT, 28); break;                                      \
    case 32: SK_MM(XT, 32); break;                                      \
    case 36: SK_MM(XT, 36); break;                                      \
    case 40: SK_MM(XT, 40); break;                                      \
    case 44: SK_MM(XT, 44); break;                                      \
    case 48: SK_MM(XT, 48); break;                                      \
    default: SK_CHECK(false, SKDSP_ERR_UNSUPPORTED, "fir_mm: %d steps", a.K4); \
    }
    switch (h->dtype) {
    case SKDSP_C64: { SK_MMK(float2) } break;
    case SKDSP_F32: { SK_MMK(float) } break;
    case SKDSP_C128: { SK_MMKD(double2) } break;
    default: { SK_MMKD(double) } break;
    }
#undef SK_MMKD
#undef SK_MMK
#undef SK_MM
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

}  // namespace skdsp
