// resample.hip -- zero-stuff upsampler, phase-selectable downsampler, planar
// <-> interleaved helpers and the synthetic-noise generator.  gfx950.
//
//   upsample    sigsys.py:3050-3053  y = hstack((x.reshape(N,1), zeros((N,L-1)))).flatten()
//   downsample  sigsys.py:3078-3083  y = x[0:floor(N/M)*M].reshape(-1,M)[:,p]
//
// Both are pure index moves (bit-exact by construction).  HBM-bound: the
// upsampler writes 16 B per lane (1 KiB per wave instruction); the downsampler
// is a strided gather (only 1/M of every fetched line is useful -- inherent).
#include "skdsp_internal.hpp"
#include <algorithm>

namespace skdsp {

template <typename T> struct Zero { __device__ static T v() { return T(0); } };
template <> struct Zero<float2> { __device__ static float2 v() { return make_float2(0.f, 0.f); } };
template <> struct Zero<double2> { __device__ static double2 v() { return make_double2(0., 0.); } };

__device__ inline float scale_v(float a, double s) { return (float)(a * (float)s); }
__device__ inline double scale_v(double a, double s) { return a * s; }
__device__ inline float2 scale_v(float2 a, double s) { return make_float2(a.x * (float)s, a.y * (float)s); }
__device__ inline double2 scale_v(double2 a, double s) { return make_double2(a.x * s, a.y * s); }

// VEC consecutive outputs per thread = 16 bytes.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upsample_kernel(const T *__restrict__ x, int64_t n_out, int L, double scale,
                                                       bool do_scale, T *__restrict__ y)
{
    const int64_t nvec = (n_out + VEC - 1) / VEC;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o0 = v * VEC;
        T out[VEC];
        // first multiple of L at or after o0
        int64_t q = o0 / L;
        int64_t r = o0 - q * L;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            T val = Zero<T>::v();
            if (r == 0 && o0 + e < n_out) {
                val = x[q];
                if (do_scale) val = scale_v(val, scale);
            }
            out[e] = val;
            if (++r == L) { r = 0; ++q; }
        }
        if (o0 + VEC <= n_out) {
            // 16-byte store, nontemporal: the stuffed signal is written once and read by the next kernel from HBM anyway
            typedef float nt4_t __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(*reinterpret_cast<const nt4_t *>(out), reinterpret_cast<nt4_t *>(y + o0));
        } else {
            for (int e = 0; e < VEC && o0 + e < n_out; ++e) y[o0 + e] = out[e];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void downsample_kernel(const T *__restrict__ x, int64_t n_out, int M, int p,
                                                         T *__restrict__ y)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_out; k += (int64_t)gridDim.x * blockDim.x)
        y[k] = x[k * M + p];
}

// Small strides (every 64-byte line of x holds kept elements): the input span of a block of outputs comes in as whole 16-byte units (full
// lines per wave instruction), the kept elements are picked out of the LDS, and the outputs leave as consecutive elements -- the strided
// gather above issues one 4 ... 16-byte request per kept element (downsample(x, 3), 2^26 complex64: 0.141 ms, 5.1 TB/s of touched bytes;
// four of them in flight per lane changed nothing).  x 16-byte aligned; OUT outputs per block, OUT * M * sizeof(T) bytes of LDS.
template <typename T, bool AHEAD>
__global__ __launch_bounds__(256) void downsample_tile_kernel(const T *__restrict__ x, int64_t n_in, int64_t n_out, int M, int p, int OUT, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) char ds_smem[];
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int PER = 16 / (int)sizeof(T);                      // elements per 16-byte unit
    constexpr int NU = 8;                                         // units per thread and block (the host keeps a block's span within 8 x 256 units)
    const int64_t nblk = (n_out + OUT - 1) / OUT;
    const int64_t u_lim = (n_in + PER - 1) / PER;                  // (the array's last unit may be partial: the caller's buffer ends there)
    // the units of block `blk` this thread brings in.  AHEAD (round 6; elements of up to 8 bytes): requested a whole block ahead -- the loads of block
    // i + 1 are in flight while block i passes through the LDS and leaves; one block at a time, a workgroup had nothing in flight between its two
    // barriers.  Same box, 2^26 samples, old -> ahead: complex64 by 3 0.1255 -> 0.1142 ms, by 2 0.1295 -> 0.1241, by 8 0.1115 -> 0.1024, float32 by 3
    // 0.0685 -> 0.0578; complex128 by 3 0.2395 -> 0.27 (16-byte elements keep the one-block form).  Nontemporal stores of the outputs: 5 - 9 % slower.
    auto request = [&](int64_t blk, v4f *r) {
        const int64_t o0 = blk * OUT;
        const int cnt = (int)(n_out - o0 < OUT ? n_out - o0 : OUT);
        const int64_t e0 = o0 * M + p, e_end = (o0 + cnt - 1) * M + p + 1;   // elements [e0, e_end) hold this block's kept ones
        const int64_t u0 = e0 / PER;
        const int nu = (int)((e_end + PER - 1) / PER - u0);
        const v4f *src = reinterpret_cast<const v4f *>(x) + u0;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int u = (int)threadIdx.x + 256 * j;
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (u < nu) {
                if ((u0 + u + 1) * PER <= n_in) v = __builtin_nontemporal_load(src + u);
                else if (u0 + u < u_lim) {                         // the ragged last unit: element by element
                    T *e = reinterpret_cast<T *>(&v);
                    for (int i = 0; i < PER; ++i)
                        if ((u0 + u) * PER + i < n_in) e[i] = x[(u0 + u) * PER + i];
                }
            }
            r[j] = v;
        }
    };
    v4f r[NU];
    int64_t blk = blockIdx.x;
    if (AHEAD && blk < nblk) request(blk, r);
    v4f *img = reinterpret_cast<v4f *>(ds_smem);
    for (; blk < nblk; blk += gridDim.x) {
        const int64_t o0 = blk * OUT;
        const int cnt = (int)(n_out - o0 < OUT ? n_out - o0 : OUT);
        const int64_t e0 = o0 * M + p;
        const int64_t u0 = e0 / PER;
        if constexpr (AHEAD) {
#pragma unroll
            for (int j = 0; j < NU; ++j) img[(int)threadIdx.x + 256 * j] = r[j];
        } else {   // (16-byte elements: unit by unit, as they arrive)
            const int64_t e_end = (o0 + cnt - 1) * M + p + 1;
            const int nu = (int)((e_end + PER - 1) / PER - u0);
            const v4f *src = reinterpret_cast<const v4f *>(x) + u0;
            for (int u = threadIdx.x; u < nu; u += 256) {
                v4f v = {0.f, 0.f, 0.f, 0.f};
                if ((u0 + u + 1) * PER <= n_in) v = __builtin_nontemporal_load(src + u);
                else if (u0 + u < u_lim) {
                    T *e = reinterpret_cast<T *>(&v);
                    for (int i = 0; i < PER; ++i)
                        if ((u0 + u) * PER + i < n_in) e[i] = x[(u0 + u) * PER + i];
                }
                img[u] = v;
            }
        }
        __syncthreads();
        if (AHEAD && blk + gridDim.x < nblk) request(blk + gridDim.x, r);
        const T *el = reinterpret_cast<const T *>(ds_smem) + (e0 - u0 * PER);
        for (int k = threadIdx.x; k < cnt; k += 256) y[o0 + k] = el[(size_t)k * M];
        __syncthreads();
    }
}
template <typename T2, typename T>
__global__ __launch_bounds__(256) void deinterleave_kernel(const T2 *__restrict__ x, int64_t n, T *__restrict__ re,
                                                           T *__restrict__ im)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        T2 v = x[k];
        re[k] = v.x;
        im[k] = v.y;
    }
}

template <typename T2, typename T>
__global__ __launch_bounds__(256) void interleave_kernel(const T *__restrict__ re, const T *__restrict__ im, int64_t n,
                                                         T2 *__restrict__ y)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        T2 v;
        v.x = re[k];
        v.y = im[k];
        y[k] = v;
    }
}

static inline int grid_for(int64_t work_items)
{
    int64_t g = (work_items + 255) / 256;
    const int64_t cap = (int64_t)ctx().num_cus * 8;  // grid-stride the rest
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int upsample_launch(const void *x, int64_t n, int L, int dtype, double scale, void *y, hipStream_t s)
{
    SK_CHECK(L >= 1, SKDSP_ERR_BADARG, "upsample: L must be >= 1 (got %d)", L);
    if (n <= 0) return SKDSP_OK;
    const int64_t n_out = n * L;
    const bool ds = scale != 1.0;
    switch (dtype) {
    case SKDSP_F32:
        hipLaunchKernelGGL((upsample_kernel<float, 4>), dim3(grid_for((n_out + 3) / 4)), dim3(256), 0, s,
                           (const float *)x, n_out, L, scale, ds, (float *)y);
        break;
    case SKDSP_C64:
        hipLaunchKernelGGL((upsample_kernel<float2, 2>), dim3(grid_for((n_out + 1) / 2)), dim3(256), 0, s,
                           (const float2 *)x, n_out, L, scale, ds, (float2 *)y);
        break;
    case SKDSP_F64:
        hipLaunchKernelGGL((upsample_kernel<double, 2>), dim3(grid_for((n_out + 1) / 2)), dim3(256), 0, s,
                           (const double *)x, n_out, L, scale, ds, (double *)y);
        break;
    case SKDSP_C128:
        hipLaunchKernelGGL((upsample_kernel<double2, 1>), dim3(grid_for(n_out)), dim3(256), 0, s, (const double2 *)x,
                           n_out, L, scale, ds, (double2 *)y);
        break;
    default:
        SK_CHECK(false, SKDSP_ERR_BADARG, "upsample: bad dtype %d", dtype);
    }
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

// y[i L + p] = src[p pitch + i]: L equally long rows woven into one signal (the second half of multirate_FIR.up through the overlap-save
// walk for large L, fir_ols.hip: its phases leave as rows first, because an element stored between elements of other rows is a write
// request of its own).  A workgroup takes I consecutive i (I a power of two): the L row pieces come in with coalesced 16-byte loads and
// are laid into LDS as rows of pitch I + 1 (conflict-free both ways), then leave as ONE contiguous run of I L elements, 16 bytes per
// lane; the (i, p) of an output position through a multiply-high by ceil(2^32 / L) (positions below I L < 2^16).
// VEC = elements per 16 bytes (1: scalar accesses -- complex128, or a destination that is not 16-byte aligned).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void interleave_kernel(const T *__restrict__ src, int64_t n, int L, int64_t pitch, int I, int log2I, unsigned magic,
                                                         T *__restrict__ y)
{
    extern __shared__ __align__(16) unsigned char smem[];
    T *buf = reinterpret_cast<T *>(smem);
    const int t = threadIdx.x;
    const int64_t nblk = (n + I - 1) / I;
    struct alignas(sizeof(T) * VEC) Pack { T e[VEC]; };
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t i0 = blk * I;
        const int cnt = (int)(n - i0 < I ? n - i0 : I);
        // loads: (row p, VEC consecutive i) per lane and step
        const int per_row = I / VEC, steps = per_row * L;
        for (int q = t; q < steps; q += 256) {
            const int p = q >> (log2I - (VEC == 4 ? 2 : VEC == 2 ? 1 : 0)), i = (q & (per_row - 1)) * VEC;
            const T *row = src + (int64_t)p * pitch + i0;
            T *dst = buf + p * (I + 1) + i;
            if (i + VEC <= cnt) {
                const Pack v = *reinterpret_cast<const Pack *>(row + i);
#pragma unroll
                for (int e = 0; e < VEC; ++e) dst[e] = v.e[e];
            } else {
                for (int e = 0; e < VEC && i + e < cnt; ++e) dst[e] = row[i + e];
            }
        }
        __syncthreads();
        T *out = y + i0 * L;
        const int total = cnt * L;
        for (int j = t * VEC; j < total; j += 256 * VEC) {
            Pack v;
            unsigned i = (unsigned)(((unsigned long long)(unsigned)j * magic) >> 32);
            unsigned p = (unsigned)j - i * (unsigned)L;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                v.e[e] = buf[p * (I + 1) + i];   // (reads past `total` in the last pack stay inside the image and are not stored)
                if (++p == (unsigned)L) { p = 0; ++i; }
            }
            if (j + VEC <= total) {
                *reinterpret_cast<Pack *>(out + j) = v;
            } else {
                for (int e = 0; e < VEC && j + e < total; ++e) out[j + e] = v.e[e];
            }
        }
        __syncthreads();
    }
}

int interleave_launch(const void *src, int64_t n, int L, int64_t pitch, int dtype, void *y, hipStream_t s)
{
    SK_CHECK(L >= 1 && L <= 4096 && pitch >= n, SKDSP_ERR_BADARG, "interleave: bad arguments (L=%d)", L);
    if (n <= 0) return SKDSP_OK;
    const size_t esz = dtype_size(dtype);
    if (L == 1) {   // one row: a copy
        SK_HIP(hipMemcpyAsync(y, src, (size_t)n * esz, hipMemcpyDeviceToDevice, s));
        return SKDSP_OK;
    }
    int I = 1024, log2I = 10;
    while (I > 4 && (size_t)(I + 1) * L * esz > 48 * 1024 - 64) { I >>= 1; --log2I; }
    SK_CHECK((size_t)(I + 1) * L * esz <= 64 * 1024, SKDSP_ERR_UNSUPPORTED, "interleave: L = %d rows do not fit the LDS", L);
    const size_t lds = (size_t)(I + 1) * L * esz + 64;
    const unsigned magic = (unsigned)((((unsigned long long)1 << 32) + L - 1) / L);
    const int64_t nblk = (n + I - 1) / I;
    const int64_t cap = (int64_t)ctx().num_cus * 8;
    const unsigned g = (unsigned)(nblk < cap ? nblk : cap);
    // 16-byte accesses need the rows (pitch a multiple of the pack, base) and the destination 16-byte aligned
    const bool wide = ((uintptr_t)y % 16 == 0) && ((uintptr_t)src % 16 == 0) && (pitch * esz) % 16 == 0 && esz < 16;
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef double v2d __attribute__((ext_vector_type(2)));
#define SK_WEAVE(T, VEC) hipLaunchKernelGGL((interleave_kernel<T, VEC>), dim3(g), dim3(256), lds, s, (const T *)src, n, L, pitch, I, log2I, magic, (T *)y)
    switch (dtype) {
    case SKDSP_F32: if (wide) SK_WEAVE(float, 4); else SK_WEAVE(float, 1); break;
    case SKDSP_C64: if (wide) SK_WEAVE(v2f, 2); else SK_WEAVE(v2f, 1); break;
    case SKDSP_F64: if (wide) SK_WEAVE(double, 2); else SK_WEAVE(double, 1); break;
    case SKDSP_C128: SK_WEAVE(v2d, 1); break;
    default: SK_CHECK(false, SKDSP_ERR_BADARG, "interleave: bad dtype %d", dtype);
    }
#undef SK_WEAVE
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

int downsample_launch(const void *x, int64_t n, int M, int p, int dtype, void *y, hipStream_t s)
{
    SK_CHECK(M >= 1, SKDSP_ERR_BADARG, "downsample: M must be >= 1 (got %d)", M);
    SK_CHECK(p >= 0 && p < M, SKDSP_ERR_BADARG, "downsample: phase p=%d out of range for M=%d", p, M);
    const int64_t n_out = n / M;
    if (n_out <= 0) return SKDSP_OK;
    // every line of x holds kept elements and x is 16-byte aligned: through the LDS (see downsample_tile_kernel)
    const size_t esz = dtype_size(dtype);
    if ((size_t)M * esz <= 64 && M > 1 && (uintptr_t)x % 16 == 0 && n_out >= 4096) {
        // (a block's span, whatever its alignment, stays within 2048 16-byte units: 8 per thread.  Measured around this point, 2^26 complex64 by 3: 32 KiB blocks
        // x 4 workgroups per CU 0.1205 ms; x 5 0.1215; 16 KiB x 8 0.1224, x 6 0.1250, x 10 0.1277; 8 KiB x 8 0.1293, x 16 0.1218)
        int OUT = (int)((32 * 1024 - 32) / ((size_t)M * esz));
        OUT = OUT / 64 * 64;   // (whole waves in the store loop)
        size_t lds = 2048 * 16;
        if (esz == 16) {       // (16-byte elements, the one-block form: blocks of whole workgroup sweeps and only their own bytes of LDS measured 3 - 5 % ahead)
            OUT = (int)(32 * 1024 / ((size_t)M * esz)) / 256 * 256;
            lds = (size_t)OUT * M * esz + 32;
        }
        const int64_t nblk = (n_out + OUT - 1) / OUT;
        const unsigned gt = (unsigned)std::min<int64_t>(nblk, (int64_t)ctx().num_cus * 4);
        const int64_t n_in = n;
#define SK_DST(T) hipLaunchKernelGGL((downsample_tile_kernel<T, (sizeof(T) < 16)>), dim3(gt), dim3(256), lds, s, (const T *)x, n_in, n_out, M, p, OUT, (T *)y)
        switch (dtype) {
        case SKDSP_F32: SK_DST(float); break;
        case SKDSP_C64: SK_DST(float2); break;
        case SKDSP_F64: SK_DST(double); break;
        case SKDSP_C128: SK_DST(double2); break;
        default: SK_CHECK(false, SKDSP_ERR_BADARG, "downsample: bad dtype %d", dtype);
        }
#undef SK_DST
        SK_HIP(hipGetLastError());
        return SKDSP_OK;
    }
    const int g = grid_for(n_out);
    switch (dtype) {
    case SKDSP_F32:
        hipLaunchKernelGGL((downsample_kernel<float>), dim3(g), dim3(256), 0, s, (const float *)x, n_out, M, p, (float *)y);
        break;
    case SKDSP_C64:
        hipLaunchKernelGGL((downsample_kernel<float2>), dim3(g), dim3(256), 0, s, (const float2 *)x, n_out, M, p, (float2 *)y);
        break;
    case SKDSP_F64:
        hipLaunchKernelGGL((downsample_kernel<double>), dim3(g), dim3(256), 0, s, (const double *)x, n_out, M, p, (double *)y);
        break;
    case SKDSP_C128:
        hipLaunchKernelGGL((downsample_kernel<double2>), dim3(g), dim3(256), 0, s, (const double2 *)x, n_out, M, p, (double2 *)y);
        break;
    default:
        SK_CHECK(false, SKDSP_ERR_BADARG, "downsample: bad dtype %d", dtype);
    }
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

int deinterleave_launch(const void *x, int64_t n, int dt, void *re, void *im, hipStream_t s)
{
    if (n <= 0) return SKDSP_OK;
    if (dt == SKDSP_C64)
        hipLaunchKernelGGL((deinterleave_kernel<float2, float>), dim3(grid_for(n)), dim3(256), 0, s, (const float2 *)x, n,
                           (float *)re, (float *)im);
    else
        hipLaunchKernelGGL((deinterleave_kernel<double2, double>), dim3(grid_for(n)), dim3(256), 0, s, (const double2 *)x,
                           n, (double *)re, (double *)im);
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

int interleave_launch(const void *re, const void *im, int64_t n, int dt, void *y, hipStream_t s)
{
    if (n <= 0) return SKDSP_OK;
    if (dt == SKDSP_C64)
        hipLaunchKernelGGL((interleave_kernel<float2, float>), dim3(grid_for(n)), dim3(256), 0, s, (const float *)re,
                           (const float *)im, n, (float2 *)y);
    else
        hipLaunchKernelGGL((interleave_kernel<double2, double>), dim3(grid_for(n)), dim3(256), 0, s, (const double *)re,
                           (const double *)im, n, (double2 *)y);
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

// ---------------------------------------------------------------- noise
// Counter-based: value depends only on (seed, global scalar index).  splitmix64
// -> two 24-bit uniforms -> Box-Muller.  Complex samples are scaled by 1/sqrt(2)
// so E|x|^2 = 1 (SURVEY.md 8d synthetic inputs).
__device__ inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ inline float2 gauss_pair(uint64_t seed, uint64_t idx)
{
    const uint64_t h = splitmix64(seed ^ (idx * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull));
    const float u1 = ((float)((h >> 40) & 0xFFFFFF) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
    const float u2 = (float)((h >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);           // [0,1)
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.28318530717958647692f * u2, &sn, &cs);
    return make_float2(r * cs, r * sn);
}

template <typename T2, typename T>
__global__ __launch_bounds__(256) void noise_complex_kernel(T2 *x, int64_t n, uint64_t seed, int64_t first)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        float2 g = gauss_pair(seed, (uint64_t)(first + k));
        T2 v;
        v.x = (T)(g.x * 0.70710678118654752440f);
        v.y = (T)(g.y * 0.70710678118654752440f);
        x[k] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void noise_real_kernel(T *x, int64_t n, uint64_t seed, int64_t first)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t gi = first + k;
        float2 g = gauss_pair(seed, (uint64_t)(gi >> 1));
        x[k] = (T)((gi & 1) ? g.y : g.x);
    }
}

int fill_noise_launch(void *x, int64_t n, int dtype, uint64_t seed, int64_t first, hipStream_t s)
{
    if (n <= 0) return SKDSP_OK;
    const int g = grid_for(n);
    switch (dtype) {
    case SKDSP_F32: hipLaunchKernelGGL((noise_real_kernel<float>), dim3(g), dim3(256), 0, s, (float *)x, n, seed, first); break;
    case SKDSP_F64: hipLaunchKernelGGL((noise_real_kernel<double>), dim3(g), dim3(256), 0, s, (double *)x, n, seed, first); break;
    case SKDSP_C64: hipLaunchKernelGGL((noise_complex_kernel<float2, float>), dim3(g), dim3(256), 0, s, (float2 *)x, n, seed, first); break;
    case SKDSP_C128: hipLaunchKernelGGL((noise_complex_kernel<double2, double>), dim3(g), dim3(256), 0, s, (double2 *)x, n, seed, first); break;
    default: SK_CHECK(false, SKDSP_ERR_BADARG, "fill_noise: bad dtype %d", dtype);
    }
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

// float32 -> float64 widening of a result vector on the device (complex = 2 scalars per sample): the
// reference's result dtype is float64/complex128, and a NumPy astype() of 2^24 complex64 on the host
// costs 15 ms against 2.4 ms of extra D2H for the already-wide copy.
__global__ __launch_bounds__(256) void widen_kernel(const float4 *__restrict__ src, int64_t n4, double *__restrict__ dst,
                                                    const float *__restrict__ tail_src, int tail)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = src[i];
        double2 *d = reinterpret_cast<double2 *>(dst + 4 * i);
        d[0] = make_double2((double)v.x, (double)v.y);
        d[1] = make_double2((double)v.z, (double)v.w);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst[4 * n4 + threadIdx.x] = (double)tail_src[threadIdx.x];
}

// y[i] += t[i] over real scalars (the partial results of a tap-partitioned FIR, capi.hip)
template <typename R>
__global__ __launch_bounds__(256) void accumulate_kernel(R *__restrict__ y, const R *__restrict__ t, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] += t[i];
}

int accumulate_launch(void *y, const void *t, int64_t nscalars, bool dbl, hipStream_t s)
{
    if (nscalars <= 0) return SKDSP_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((nscalars + 255) / 256, 8192);
    if (dbl) hipLaunchKernelGGL(accumulate_kernel<double>, dim3(blocks), dim3(256), 0, s, (double *)y, (const double *)t, nscalars);
    else hipLaunchKernelGGL(accumulate_kernel<float>, dim3(blocks), dim3(256), 0, s, (float *)y, (const float *)t, nscalars);
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

// element-wise float32 <-> float64 between device vectors of any (element) alignment: the float64 detour of float32 cascades whose
// sections cannot be grouped in float32 (iir_scan.hip)
template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_kernel(const S *__restrict__ src, D *__restrict__ dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (D)src[i];
}

int convert_launch(const void *src, void *dst, int64_t nscalars, bool to_double, hipStream_t s)
{
    if (nscalars <= 0) return SKDSP_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((nscalars + 255) / 256, 8192);
    if (to_double) hipLaunchKernelGGL((convert_kernel<float, double>), dim3(blocks), dim3(256), 0, s, (const float *)src, (double *)dst, nscalars);
    else hipLaunchKernelGGL((convert_kernel<double, float>), dim3(blocks), dim3(256), 0, s, (const double *)src, (float *)dst, nscalars);
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

int widen_launch(const void *src, int64_t nscalars, void *dst, hipStream_t s)
{
    if (nscalars <= 0) return SKDSP_OK;
    const int64_t n4 = nscalars / 4;
    const int tail = (int)(nscalars - 4 * n4);
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(widen_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float4 *)src, n4, (double *)dst,
                       (const float *)src + 4 * n4, tail);
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

// Zero-stuffing of an interleaved complex vector straight into the two real planes the IIR scan works
// on (re[o] = scale * Re x[o / L] for o % L == 0, else 0): saves writing and re-reading the stuffed
// interleaved vector in front of rate_change.up / multirate_IIR.up on complex data.
template <typename C, typename R, int VEC>
__global__ __launch_bounds__(256) void upsample_planes_kernel(const C *__restrict__ x, int64_t n_out, int L, R scale,
                                                              R *__restrict__ re, R *__restrict__ im)
{
    const int64_t nvec = (n_out + VEC - 1) / VEC;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o0 = v * VEC;
        int64_t q = o0 / L;
        int64_t r = o0 - q * L;
        R a[VEC], b[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            a[e] = R(0);
            b[e] = R(0);
            if (r == 0 && o0 + e < n_out) {
                const C val = x[q];
                a[e] = val.x * scale;
                b[e] = val.y * scale;
            }
            if (++r == L) { r = 0; ++q; }
        }
        if (o0 + VEC <= n_out) {
            *reinterpret_cast<float4 *>(re + o0) = *reinterpret_cast<const float4 *>(a);
            *reinterpret_cast<float4 *>(im + o0) = *reinterpret_cast<const float4 *>(b);
        } else {
            for (int e = 0; e < VEC && o0 + e < n_out; ++e) {
                re[o0 + e] = a[e];
                im[o0 + e] = b[e];
            }
        }
    }
}

int upsample_planes_launch(const void *x, int64_t n, int L, int dtype, double scale, void *re, void *im, hipStream_t s)
{
    SK_CHECK(L >= 1 && dtype_complex(dtype), SKDSP_ERR_BADARG, "upsample_planes: needs a complex dtype and L >= 1");
    const int64_t n_out = n * L;
    if (n_out <= 0) return SKDSP_OK;
    if (dtype == SKDSP_C64)
        hipLaunchKernelGGL((upsample_planes_kernel<float2, float, 4>), dim3(grid_for((n_out + 3) / 4)), dim3(256), 0, s,
                           (const float2 *)x, n_out, L, (float)scale, (float *)re, (float *)im);
    else
        hipLaunchKernelGGL((upsample_planes_kernel<double2, double, 2>), dim3(grid_for((n_out + 1) / 2)), dim3(256), 0, s,
                           (const double2 *)x, n_out, L, scale, (double *)re, (double *)im);
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

}  // namespace skdsp
