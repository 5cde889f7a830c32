// fir_up2k.hip -- multirate_FIR.up (multirate_helper.py:112-118, lfilter(b, [1], L * upsample(x, L))) in the frequency
// domain for MANY phases: one workgroup per INPUT tile, the tile loaded and transformed once, the L phase filters
//   y[i L + p] = L sum_t b[p + L t] x[i - t]
// as L pointwise products + inverse transforms of that one spectrum -- and a thread keeps the results of ALL phases of its
// eight samples (up to 12 complex64 phases / 24 float32 phases = 192 registers), so that a sample's whole output row leaves
// in one burst.  Measured on this board (tools/ubench_strided_store.hip): rows of 64 ... 512 bytes written as 32-byte pieces
// back to back run at 5 TB/s, the same pieces written a sweep apart at 1.2 - 1.6 TB/s -- which is what the 4096-point kernel
// with four phases per thread (fir_up4k.hip) pays from L = 5 on (0.34 ms per 2^26 outputs at L = 8, 0.41 at L = 12 against
// 0.12 for its transforms alone), and what the walk over (tile, phase) pairs before it paid at every L.
//
// Tile: 2048 complex64 points, 256 threads x 8 points (ols2k_core.hpp); V = 2048 - OV inputs -> V L outputs per tile, OV =
// taps per phase - 1 rounded up to 64 (a wave's 64 lanes own 64 consecutive samples of each of the tile's 8 blocks: whole
// (block, wave) pairs are skipped).  float32 signals with real taps run their phases in PAIRS (x * (h_2q + i h_2q+1) = y_2q
// + i y_2q+1: one complex pass over the real tile yields the interleaved pair as one 8-byte element).
// The rows of a (block, wave) pair are 64 consecutive rows of y = one contiguous run: they go through a wave-private
// staging image in the LDS and leave as consecutive 16 bytes per lane, 1 KiB per store instruction.
// Algorithmic bytes: 8 B x (n + n L) complex64, 4 B x (n + n L) float32.
#include "skdsp_internal.hpp"
#include "ols2k_tables.hpp"

namespace skdsp {

using namespace ols2k;
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));

struct Up2kPlan {
    int L = 0, T = 0, ov = 0, V = 0, passes = 0;
    bool pairs = false;
    float2 *tw1 = nullptr, *tw2 = nullptr, *tw3 = nullptr;
    float4 *Hp = nullptr;
};

struct Up2kArgs {
    const void *x;
    void *y;
    int64_t n, n_hist;       // input samples; valid history in front of x[0]
    const float2 *tw1, *tw2, *tw3;
    const float4 *Hp;        // passes x 1024 float4
    int ov, V;               // ov a multiple of 64
    int passes;              // complex passes per tile: L (complex64) or ceil(L / 2) (float32: two phases per pass)
    int row_bytes;           // bytes of one input sample's L outputs
    int odd_tail;            // float32, odd L: the last pass carries ONE phase (4 bytes)
    int aligned;             // x and y element-aligned
    int staged;              // rows leave through the staging image (needs an even number of passes and no tail)
    unsigned upr_magic;      // staged: ceil(65536 / (row_bytes / 16)): lane / units-per-row by multiply-high
    int64_t ntiles;
    CarefulFir cf;           // the filter as the exact path of a poisoned tile reads it (careful.hpp)
};

__device__ __forceinline__ bool up2k_interior(const Up2kArgs &A, int64_t tile)
{
    const int64_t in0 = tile * A.V - A.ov;
    return A.aligned && in0 >= -A.n_hist && in0 + k2N <= A.n;
}
// x[in0 + 256 m + t] -> v[m]  (v[2 a + e] = x[512 a + 256 e + t]: ols2k_core.hpp).  XR: a float32 signal into the real parts
// (the imaginary parts are set where the tile is transformed).
template <bool XR> __device__ __forceinline__ void up2k_load_interior(const Up2kArgs &A, int64_t tile, int t, cf *v)
{
    const int64_t in0 = tile * A.V - A.ov;
    int tt = t;   // (opaque copy: the addresses are rebuilt per tile instead of living in registers across the tile loop)
    asm volatile("" : "+v"(tt));
    if (XR) {
        const float *xp = reinterpret_cast<const float *>(A.x) + in0;
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m].x = __builtin_nontemporal_load(xp + (unsigned)(m * 256 + tt));
    } else {
        const v2f_t *xp = reinterpret_cast<const v2f_t *>(A.x) + in0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const v2f_t r = __builtin_nontemporal_load(xp + (unsigned)(m * 256 + tt));
            v[m] = make_float2(r.x, r.y);
        }
    }
}
// (out of line, through a small array in scratch: the first and last tiles of a signal only; arguments by value, so that the
// kernel's argument block stays in scalar registers)
template <bool XR> __device__ __noinline__ void up2k_load_edge(const void *x, int64_t in0, int64_t n_hist, int64_t n, int t, cf *v)
{
    for (int m = 0; m < 8; ++m) {
        const int64_t g = in0 + 256 * m + t;
        cf val = make_float2(0.f, 0.f);
        if (g >= -n_hist && g < n) {
            if (XR) val.x = reinterpret_cast<const float *>(x)[g];
            else val = reinterpret_cast<const cf *>(x)[g];
        }
        v[m] = val;
    }
}
template <bool XR> __device__ __forceinline__ void up2k_settle_x(const cf *v)
{
    if constexpr (XR)
        asm volatile("" ::"v"(v[0].x), "v"(v[1].x), "v"(v[2].x), "v"(v[3].x), "v"(v[4].x), "v"(v[5].x), "v"(v[6].x), "v"(v[7].x) : "memory");
    else
        asm volatile("" ::"v"(v[0].x), "v"(v[0].y), "v"(v[1].x), "v"(v[1].y), "v"(v[2].x), "v"(v[2].y), "v"(v[3].x), "v"(v[3].y), "v"(v[4].x), "v"(v[4].y),
                     "v"(v[5].x), "v"(v[5].y), "v"(v[6].x), "v"(v[6].y), "v"(v[7].x), "v"(v[7].y)
                     : "memory");
}
// "these are the results, in these registers, now": without it hipcc carries a finished pass in a form of its own (24 live registers per
// pass instead of 16 -- 142 / 236 / 256 + 68 spilled for 4 / 8 / 12 passes per thread; with it 122 / 184 / 250 and no spill)
__device__ __forceinline__ void up2k_pin(cf *v)
{
    asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[3].x), "+v"(v[3].y), "+v"(v[4].x), "+v"(v[4].y),
                 "+v"(v[5].x), "+v"(v[5].y), "+v"(v[6].x), "+v"(v[6].y), "+v"(v[7].x), "+v"(v[7].y));
}
__device__ __forceinline__ float4 up2k_vld(const volatile float4 *p)
{
    float4 r;
    r.x = p->x; r.y = p->y; r.z = p->z; r.w = p->w;
    return r;
}

// float4 units per staged row: the row (8 bytes per pass) plus one unit of padding where the row length in units is even, so
// that the 8 lanes a ds_write_b128 serves per cycle hit 8 different 16-byte bank groups (an odd unit stride)
template <int PH> struct Up2kStage {
    static constexpr int kRowUnits = (PH / 2) % 2 == 0 ? PH / 2 + 1 : PH / 2;
    static constexpr int kWaveUnits = 64 * kRowUnits;
};

// The rows of (block m, this wave) when the launch holds ALL passes of a row per thread (cnt = passes, even, no tail): 64 consecutive
// rows of y = one contiguous run.  A lane writes its row into the staging image, then the wave copies the image out as consecutive
// 16-byte units (unit u = lane + 64 k of the run sits in the image at row u / upr, piece u % upr): uniform 64-bit base + one 32-bit
// lane offset per store.
template <int PH> __device__ __forceinline__ void up2k_store_staged(const Up2kArgs &A, int64_t tile, int cnt, int t, const cf *out, float4 *stage)
{
    constexpr int RU = Up2kStage<PH>::kRowUnits;
    int tt = t;   // (opaque copy: nothing of the store addressing is hoisted out of the tile loop)
    asm volatile("" : "+v"(tt));
    const int lane = tt & 63, wv = tt >> 6;
    const int upr = cnt >> 1;                                          // 16-byte units per row
    const int r0 = (int)(((unsigned)lane * A.upr_magic) >> 16);      // lane / upr
    const int j0 = lane - r0 * upr;
    const int q64 = 64 / upr, m64 = 64 - q64 * upr;                   // 64 = q64 upr + m64
    const int64_t out0 = tile * A.V;
    const int64_t left = A.n - out0;                                   // rows of this tile that exist
    const unsigned voff = 16u * (unsigned)lane;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int s0 = 256 * m + 64 * wv - A.ov;                      // first tile-local output row of (block, wave); uniform
        if (s0 < 0 || s0 >= left) continue;
        const int rows = left - s0 < 64 ? (int)(left - s0) : 64;
#pragma unroll
        for (int j = 0; j < PH / 2; ++j)
            if (j < upr) stage[lane * RU + j] = make_float4(out[8 * (2 * j) + m].x, out[8 * (2 * j) + m].y, out[8 * (2 * j + 1) + m].x, out[8 * (2 * j + 1) + m].y);
        char *run = reinterpret_cast<char *>(A.y) + (out0 + s0) * A.row_bytes;   // uniform
        int r = r0, j = j0;
#pragma unroll
        for (int k = 0; k < PH / 2; ++k) {
            if (k < upr) {
                const float4 w = stage[r * RU + j];
                if (r < rows) *reinterpret_cast<v4f_t *>(run + 1024 * k + voff) = v4f_t{w.x, w.y, w.z, w.w};
                r += q64; j += m64;
                if (j >= upr) { j -= upr; r += 1; }
            }
        }
    }
}

// The same rows, each lane its own (any number of passes; TAIL: the last pass is one float32 phase, 4 bytes): 16 bytes per pair of
// passes.  The pieces of a row still leave back to back.
template <int PH, bool TAIL> __device__ __forceinline__ void up2k_store_direct(const Up2kArgs &A, int64_t tile, int g0, int cnt, int t, const cf *out)
{
    int tt = t;
    asm volatile("" : "+v"(tt));
    const int64_t out0 = tile * A.V;
    const int64_t left = A.n - out0;
    const int wv = tt >> 6;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int s0 = 256 * m + 64 * wv - A.ov;   // uniform
        if (s0 < 0 || s0 >= left) continue;
        const int s = s0 + (tt & 63);
        if (s >= left) continue;
        char *p = reinterpret_cast<char *>(A.y) + (out0 + s) * A.row_bytes + 8 * g0;
#pragma unroll
        for (int q = 0; q < PH; q += 2) {
            if (q + 1 < cnt && !(TAIL && q + 2 == cnt)) {
                *reinterpret_cast<v4f_t *>(p + 8 * q) = v4f_t{out[8 * q + m].x, out[8 * q + m].y, out[8 * (q + 1) + m].x, out[8 * (q + 1) + m].y};
            } else if (q + 1 < cnt) {   // (TAIL: the pair's second pass is the single last phase)
                *reinterpret_cast<v2f_t *>(p + 8 * q) = v2f_t{out[8 * q + m].x, out[8 * q + m].y};
                *reinterpret_cast<float *>(p + 8 * q + 8) = out[8 * (q + 1) + m].x;
            } else if (q < cnt) {
                if (TAIL) *reinterpret_cast<float *>(p + 8 * q) = out[8 * q + m].x;
                else *reinterpret_cast<v2f_t *>(p + 8 * q) = v2f_t{out[8 * q + m].x, out[8 * q + m].y};
            }
        }
    }
}

// A poisoned tile (see fir_up4k.hip): the thread recomputes the rows it stored -- tile-local rows 256 m + t - ov -- by the reference's own sum.
template <bool XR> __device__ __forceinline__ void up2k_careful_rows(const void *x, void *y, int64_t n, int64_t n_hist, const CarefulFir cf, int L, int64_t out0, int ov, int t)
{
#pragma unroll 1
    for (int m = 0; m < 8; ++m) {
        const int s = 256 * m + t - ov;
        if (s < 0) continue;
        if (out0 + s >= n) break;
        careful_up_row<XR>(x, y, n_hist, cf, L, out0 + s);
    }
}

// Persistent: 2 workgroups per CU walk the input tiles (XCD-contiguous runs per round, like ols_tile_kernel).  PH = passes whose
// results a thread holds before it stores (its register budget: 16 per pass).  Every pass works in place in the registers of its
// result: H product -> inverse pass 4 -> (LDS) -> inverse passes 3, 2 (arrays of their own, alive only while the result's
// registers are dead) -> inverse pass 1.  The next tile's samples are requested behind the H product of the tile's last pass
// (into the registers of the spectrum, dead by then) and waited for in front of the stores: vmcnt retires in order, a wait
// behind a store burst is a wait for its acknowledgements.
// ZL (twelve passes per thread): the spectrum waits in thread-private LDS slots instead of 16 registers and the next tile is not
// requested ahead -- with 192 result registers, the landing registers of a pass's LDS reads and its twiddles, there is no room
// for either (199 - 221 spilled registers otherwise, whatever the instruction scheduler).
template <bool XR, int PH, bool STAGED> __global__ __launch_bounds__(256, 2) void up2k_kernel(Up2kArgs A)
{
    constexpr bool ZL = PH > 8;
    __shared__ cf img[kImgUnits];
    __shared__ cf tw1l[kTw1Units], tw2l[kTw2Units], tw3l[kTw3Units];
    __shared__ float4 stage[STAGED ? 4 * Up2kStage<PH>::kWaveUnits : 1];
    __shared__ cf zl[ZL ? 8 * 256 : 1];   // [slot][thread]
    __shared__ unsigned long long up_noted;   // poisoned tiles, by walk step (careful.hpp)
    const int t = threadIdx.x;
    if (t == 0) up_noted = 0;
    {
        tw1l[t] = A.tw1[t]; tw1l[256 + t] = A.tw1[256 + t]; tw1l[512 + t] = A.tw1[512 + t];
        tw2l[t] = A.tw2[t]; tw2l[256 + t] = A.tw2[256 + t];
        if (t < kTw3Units) tw3l[t] = A.tw3[t];
    }
    __syncthreads();
    int64_t tile = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
    auto tile_first = [&]() -> int64_t { return (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x; };
    cf Z[8];               // the tile's samples, then its spectrum, then (behind the last H product) the next tile's samples
    bool have_x = false;   // Z holds the samples of `tile` (requested a pass ahead: interior tiles only)
    for (; tile < A.ntiles; tile += gridDim.x) {
        const bool has_next = tile + gridDim.x < A.ntiles;
        const bool pre_next = !ZL && has_next && up2k_interior(A, tile + gridDim.x);
        if (!have_x) {   // the first tile of this workgroup, and tiles at the ends of the signal (guarded accesses)
            if (up2k_interior(A, tile)) {
                up2k_load_interior<XR>(A, tile, t, Z);
            } else {
                cf e[8];
                up2k_load_edge<XR>(A.x, tile * A.V - A.ov, A.n_hist, A.n, t, e);
#pragma unroll
                for (int m = 0; m < 8; ++m) Z[m] = e[m];
            }
        }
        have_x = pre_next;
        if constexpr (XR) {
#pragma unroll
            for (int m = 0; m < 8; ++m) Z[m].y = 0.f;
        }
        fwd_pass1(t, Z, tw1l, img);
        __syncthreads();
        fwd_pass2(t, tw2l, img);   // (from here to the H product: wave-local -- a wave owns the region of its k1)
        fwd_pass3(t, tw3l, img);
        fwd_pass4(t, img, Z);
        if constexpr (ZL) {
#pragma unroll
            for (int k = 0; k < 8; ++k) zl[k * 256 + t] = Z[k];
        }
        bool poisoned = false;
        for (int g0 = 0; g0 < A.passes; g0 += PH) {
            const int cnt = A.passes - g0 < PH ? A.passes - g0 : PH;
            const bool last_group = g0 + cnt == A.passes;
            cf out[PH * 8];
            static_for<0, PH>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                if (q < cnt) {
                    float4 hh[4];
                    {
                        int tt = t;
                        asm volatile("" : "+v"(tt));
                        const volatile float4 *hp = reinterpret_cast<const volatile float4 *>(A.Hp) + (size_t)(g0 + q) * 1024;
#pragma unroll
                        for (int k = 0; k < 4; ++k) hh[k] = up2k_vld(hp + (unsigned)(k * 256 + tt));
                    }
                    if constexpr (ZL) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) out[8 * q + k] = zl[k * 256 + t];
                        mul_H(hh, out + 8 * q, out + 8 * q);
                    } else {
                        mul_H(hh, Z, out + 8 * q);
                        if (q == cnt - 1 && last_group && pre_next)   // the tile's last pass: the spectrum is dead, the next tile's samples land in its registers
                            up2k_load_interior<XR>(A, tile + gridDim.x, t, Z);
                    }
                    inv_pass4(t, img, out + 8 * q);
                    inv_pass3<ZL>(t, tw3l, img);
                    inv_pass2<ZL>(t, tw2l, img);
                    __syncthreads();
                    inv_pass1(t, tw1l, img, out + 8 * q);
                    __syncthreads();   // every wave has read the image before the next pass (or tile) overwrites it
                    up2k_pin(out + 8 * q);
                }
            });
            if (!ZL && last_group && pre_next) up2k_settle_x<XR>(Z);
            poisoned |= not_finite(out[7].x) | not_finite(out[7].y);
            if constexpr (STAGED) {
                up2k_store_staged<PH>(A, tile, cnt, t, out, stage + Up2kStage<PH>::kWaveUnits * (t >> 6));
            } else if constexpr (XR) {
                if (A.odd_tail && last_group) up2k_store_direct<PH, true>(A, tile, g0, cnt, t, out);
                else up2k_store_direct<PH, false>(A, tile, g0, cnt, t, out);
            } else {
                up2k_store_direct<PH, false>(A, tile, g0, cnt, t, out);
            }
        }
        if (__builtin_expect(__any(poisoned), 0)) careful_note(&up_noted, (tile - tile_first()) / gridDim.x);
    }
    const unsigned long long noted = careful_noted(&up_noted);
    if (__builtin_expect(noted != 0, 0)) {
        int64_t k = 0;
        for (int64_t tl = tile_first(); tl < A.ntiles; tl += gridDim.x, ++k)
            if (careful_step_noted(noted, k)) up2k_careful_rows<XR>(A.x, A.y, A.n, A.n_hist, A.cf, A.row_bytes / (XR ? 4 : 8), tl * A.V, A.ov, t);
    }
}

struct Up2kPlanList { std::vector<Up2kPlan *> plans; };

static void up2k_free_plan(Up2kPlan *p)
{
    if (!p) return;
    if (p->tw1) (void)hipFree(p->tw1);
    if (p->tw2) (void)hipFree(p->tw2);
    if (p->tw3) (void)hipFree(p->tw3);
    if (p->Hp) (void)hipFree(p->Hp);
    delete p;
}

void fir_up2k_free(void *list)
{
    Up2kPlanList *l = static_cast<Up2kPlanList *>(list);
    if (!l) return;
    for (Up2kPlan *p : l->plans) up2k_free_plan(p);
    delete l;
}

// complex64 (any taps) or float32 with real taps; per phase at most 1025 taps (half a tile of overlap)
bool fir_up2k_supported(const FirHandle *h, int L)
{
    // (a plan holds one 32 / 16 KiB table per pass, built on first use under the handle's lock: 256 passes are 8 / 4 MiB and a few ms of host transforms;
    // beyond that the walk over (tile, phase) pairs and the polyphase kernels serve the call)
    if (L < 2 || L > 256) return false;
    const int T = up_taps_per_phase(h->ntaps, L);
    if (T - 1 > 1024) return false;
    return h->dtype == SKDSP_C64 || (h->dtype == SKDSP_F32 && !h->taps_complex);
}

static int up2k_plan(FirHandle *h, int L, Up2kPlan **out)
{
    if (!h->up2k) h->up2k = new Up2kPlanList();
    Up2kPlanList *l = static_cast<Up2kPlanList *>(h->up2k);
    for (Up2kPlan *p : l->plans)
        if (p->L == L) { *out = p; return SKDSP_OK; }
    Up2kPlan *p = new Up2kPlan();
    p->L = L;
    p->pairs = h->dtype == SKDSP_F32;
    p->T = up_taps_per_phase(h->ntaps, L);
    p->ov = ((p->T - 1 + 63) / 64) * 64;
    if (p->ov == 0) p->ov = 64;
    p->V = k2N - p->ov;
    p->passes = up_passes(L, p->pairs);
    std::vector<float2> tw1, tw2, tw3;
    std::vector<float4> Hp;
    make_tw1(tw1);
    make_tw2(tw2);
    make_tw3(tw3);
    make_up_tables(h->taps_host.data(), h->ntaps, h->taps_complex ? 2 : 1, L, p->pairs, Hp);
    hipError_t e;
    if ((e = hipMalloc((void **)&p->tw1, tw1.size() * sizeof(float2))) != hipSuccess ||
        (e = hipMalloc((void **)&p->tw2, tw2.size() * sizeof(float2))) != hipSuccess ||
        (e = hipMalloc((void **)&p->tw3, tw3.size() * sizeof(float2))) != hipSuccess ||
        (e = hipMalloc((void **)&p->Hp, Hp.size() * sizeof(float4))) != hipSuccess ||
        (e = hipMemcpy(p->tw1, tw1.data(), tw1.size() * sizeof(float2), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->tw2, tw2.data(), tw2.size() * sizeof(float2), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->tw3, tw3.data(), tw3.size() * sizeof(float2), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->Hp, Hp.data(), Hp.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess) {
        up2k_free_plan(p);
        return hip_fail(e, "up2k tables", __FILE__, __LINE__);
    }
    l->plans.push_back(p);
    *out = p;
    return SKDSP_OK;
}

int fir_up2k_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, int L, void *y, hipStream_t s)
{
    note_path("fir_up2k");
    if (n <= 0) return SKDSP_OK;
    SK_CHECK(fir_up2k_supported(h, L), SKDSP_ERR_UNSUPPORTED, "fir_up2k: needs complex64 (or float32 with real taps), 2 <= L <= 4096, at most 1025 taps per phase");
    Up2kPlan *p = nullptr;
    int rc = up2k_plan(h, L, &p);
    if (rc) return rc;
    const int esz = h->dtype == SKDSP_F32 ? 4 : 8;
    Up2kArgs A;
    A.x = x; A.y = y; A.n = n; A.n_hist = n_hist;
    A.tw1 = p->tw1; A.tw2 = p->tw2; A.tw3 = p->tw3; A.Hp = p->Hp;
    A.ov = p->ov; A.V = p->V;
    A.passes = p->passes;
    A.row_bytes = L * esz;
    A.odd_tail = p->pairs && (L & 1);
    A.aligned = ((((uintptr_t)x) | ((uintptr_t)y)) & (esz - 1)) == 0;
    A.ntiles = (n + p->V - 1) / p->V;
    if ((rc = fir_careful(h, &A.cf))) return rc;
    SK_CHECK(A.ntiles < (int64_t)1 << 31, SKDSP_ERR_BADARG, "fir_up2k: too many tiles");
    // passes held per thread: the smallest instantiation that takes the row in ONE group; longer rows go in groups of 12 (float32
    // and complex64 alike: 96-byte pieces)
    const int PH = p->passes <= 4 ? 4 : (p->passes <= 8 ? 8 : 12);
    // the staging image takes whole rows with an even number of passes and no 4-byte tail
    A.staged = opt().fir_up4k_staged && !A.odd_tail && p->passes % 2 == 0 && p->passes <= PH;
    {
        const int cnt0 = p->passes < PH ? p->passes : PH;
        const int upr = cnt0 / 2 > 0 ? cnt0 / 2 : 1;
        A.upr_magic = (unsigned)((65536 + upr - 1) / upr);
    }
    int64_t grid = 2 * (int64_t)ctx().num_cus;
    const int reserve_wgs = opt().ols_reserve;
    if (reserve_wgs > 0 && grid >= 4 * (int64_t)reserve_wgs) grid -= reserve_wgs;
    if (grid > A.ntiles) grid = A.ntiles;
    const dim3 g((unsigned)grid), b(256);
    auto launch = [&](auto xr, auto ph) {
        constexpr bool X = decltype(xr)::value;
        constexpr int P = decltype(ph)::value;
        if (A.staged) hipLaunchKernelGGL((up2k_kernel<X, P, true>), g, b, 0, s, A);
        else hipLaunchKernelGGL((up2k_kernel<X, P, false>), g, b, 0, s, A);
    };
    auto by_ph = [&](auto xr) {
        if (PH == 4) launch(xr, std::integral_constant<int, 4>{});
        else if (PH == 8) launch(xr, std::integral_constant<int, 8>{});
        else launch(xr, std::integral_constant<int, 12>{});
    };
    if (p->pairs) by_ph(std::true_type{}); else by_ph(std::false_type{});
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

}  // namespace skdsp
