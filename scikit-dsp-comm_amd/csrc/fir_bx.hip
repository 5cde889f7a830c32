// fir_bx.hip -- direct / polyphase FIR of float32 or complex64 signals with real taps as a Toeplitz matrix product on the
// fp16 matrix pipe, at float32 accuracy: every operand is split into TWO fp16 pieces (x = x1 + x2: 22 - 23 of float32's 24
// mantissa bits, round-to-nearest pieces, residual <= 2^-23 |x|: half an ulp more than float32 itself carries), and the three
// partial products down to 2^-11 of the leading one are formed with v_mfma_f32_16x16x32_f16 (fp32 accumulate; the two small
// products in their own accumulator); x2 h2 (2^-22) is not formed.  fp16's exponent range is narrow, so every WINDOW of the signal
// is multiplied by the power of two that puts its largest magnitude at 2^14 before it is split, its outputs by the inverse, and
// the taps are scaled once on the host: exact, and scale-invariant over the whole float32 range
// (tests/test_gpu_parity.py::test_matrix_pipe_path_is_scale_invariant).  The second piece of every operand is carried lifted by 2^11
// (kBxLift), so a sample keeps its 22 bits down to 2^-29 of its window's largest magnitude and an absolute accuracy of 2^-51 of that
// largest below it.
// Until round 4 the pieces were three bf16 (exact 24 bits, 8-bit exponents, no scaling) and the products six: the kernels ran at the
// board's power cap with the matrix pipe ~60 % busy, and half the products were half the joules -- config 3 0.313 -> 0.238 ms, 127 taps
// float32 0.135 -> 0.111, same error statistics on the coherent-input suite (tests/test_gpu_adversarial.py; rel-L2 1.3e-7).
//
// Serves the same reference calls as fir_mm.hip / fir_direct.hip (multirate_helper.py:104-127 and
// downsample(up(x,L),M)):   y[m] = L * sum_t b[phi_c + L t] * x[i_c + q s - t],
//     m = c + L' s,  c = m mod L',  L' = L/gcd, q = M/gcd,  phi_c = (c M) mod L, i_c = (c M) div L.
//
// Output ROWS r = L' ds + c are DS consecutive slots of all L' classes (RS = L' DS rows, RT = ceil(RS/16)
// row tiles), a COLUMN N is a slot block, so   m = RS N + r   and the input index is
//     q DS N + U0 - u,   lag u = t + U0 - i_c - q ds >= 0   (independent of N):
//     Y[RS x N] = A[RS x K] * W[K x N],  A[r][u] = L b[phi_c + L (u - U0 + i_c + q ds)],  W[u][N] = x[q DS N + U0 - u].
// The instruction wants 8 consecutive lags per lane.  With k' = K-1-u ascending in x, lane (column n,
// group j) reads the 8 window elements  q DS n + 32 kb + 8 j + (0..7)  as ONE 16-byte LDS read per fp16
// piece -- aligned whenever q DS is a multiple of 8, which fixes DS (and with it RS: 32 rows for L/M = 4/3,
// 16 for a plain filter, 96 for L = 12).  The taps (two fp16 pieces per row tile and 32-lag block) stay in
// registers for the whole launch; the window is split once, while it is staged, into 2 (4: re, im) fp16
// planes in LDS.  The MFMA is issued with the window fragment as its A operand and the taps as B, i.e. the
// tile comes out transposed (col = lane & 15 = row of the tile, the four registers = four columns), so that
// the 16 lanes of a group store 16 consecutive outputs.
#include "skdsp_internal.hpp"
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>


namespace skdsp {

typedef float v4f_bx __attribute__((ext_vector_type(4)));
typedef float v2f_bx __attribute__((ext_vector_type(2)));
typedef _Float16 v8h_bx __attribute__((ext_vector_type(8)));

constexpr int kBxUnitsC = 512;   // 8-sample units of one complex64 window (2 per thread: 32 prefetch VGPRs)
constexpr int kBxUnitsR = 1024;  // float32: 4 per thread, the same 32 VGPRs
// ... of the lag-split kernels (KSP = 4): their registers and the smaller fp16 planes leave room for windows of several column tiles
// where 16 columns are already thousands of samples (56 / 64 KiB of planes + the partial tiles: still two workgroups per CU)
constexpr int kBxUnitsCK = 896, kBxUnitsRK = 2048;
static constexpr int bx_units(bool cplx, bool ksp) { return ksp ? (cplx ? kBxUnitsCK : kBxUnitsRK) : (cplx ? kBxUnitsC : kBxUnitsR); }

struct BxArgs {
    int64_t n, n_hist, n_out;
    int q_ds;  // q * DS: input samples per column (multiple of 8)
    int RS;    // rows in use = L' * DS
    int U0;    // lag offset
    int NS;    // columns per workgroup (multiple of 16)
    int win;   // staged samples per workgroup (multiple of 8) >= q_ds * (NS - 1) + 32 KB
    // 1: a tile takes every second column of a 32-column span (tile 2 h: the even, 2 h + 1: the odd ones), in the row
    // order 0-3 -> 0-3, 12-15 -> 4-7, 4-11 -> 8-15 of the 16 taken.  Lane (row r, group j) reads the 16-byte unit
    // s col(r) + j + 4 kb of a plane (s = q_ds / 8 units per column), and ds_read_b128 serves the lanes in groups
    // {rows 0-3, 12-15 of group j} + {rows 4-11 of group j + 1}.  With 16 CONSECUTIVE columns an odd s makes two of
    // those 16 units share a bank whatever the row order (the two row sets must hit the same eight residues of
    // s col mod 16 shifted by one): 2-way conflicts on a third of the reads, 42 % of the LDS cycles of L/M = 4/3
    // (s = 3).  With every second column both row sets hit the eight EVEN (odd) residues exactly once: none.
    int eo;
    // pad_s > 0 (s = q_ds / 8 a multiple of 4: decimators): one pad unit behind every s units of a plane.  Unpadded, the 16 columns of a
    // fragment read start s units apart -- s = 4 / 8 / 16 / 24 puts them on 4 / 2 / 1 / 2 of the 16 bank residues (4- to 16-way conflicts
    // on every read: M = 8, 512 taps, complex64 ran at 0.30 ms per 2^26 inputs); padded, the stride is s + 1 (odd).
    int pad_s;
    unsigned pad_magic;   // ceil(2^32 / s)
    float tap_inv;        // 2^-te: the taps of the table are L b 2^te (scaled into the fp16 range on the host)
    int L, M;             // of the call (the exact path of a window this kernel cannot compute: careful.hpp)

    CarefulFir cf;
};

constexpr int kBxPx = 2;          // fp16 pieces of a signal sample
constexpr int kBxPh = 2;          // fp16 pieces of a tap
// The SECOND piece of every operand is the residual times 2^11 (kBxLift): fp16's floor is 2^-24, and a residual is 2^-11 of its first piece at
// most, so unlifted it fell off that floor for every sample 2^-25 or more below its window's largest -- lifted, a sample keeps its full 22 bits
// down to 2^-29 of the window's largest and something of itself down to 2^-50 (a 1e12 glitch leaves the rest of its window accurate to 5e-4
// of ITS level, not zero).  Both small products (h2 x1, h1 x2) carry the same 2^11, so their accumulator is scaled back once, at the store.
constexpr int kBxLift = 11;

// (a, b), already scaled into the fp16 range -> two packed fp16 pairs, a in the low half: a = a1 + 2^-11 a2 to 2^-23 |a|.
// Round-to-nearest pieces (v_cvt_pk_f16_f32), so the residual is at most half an ulp of the first piece (<= 8 -> <= 2^14 lifted).
// The residuals are formed with SCALAR instructions on purpose: v_pk_add_f32 runs on the datapath the matrix pipe uses, so while the
// other workgroup of the CU is in its MFMA phase a packed split does not advance at all (round 3: the split of one workgroup ended
// ~240 clocks after the partner's last MFMA, every window, which is what locked the two workgroups of a CU in phase).
__device__ __forceinline__ void bx_split2(float a, float b, unsigned (&p)[kBxPx])
{
    auto cvt = [](float lo, float hi) -> unsigned {
        unsigned r;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
        return r;
    };
    // (x - (float)piece.lo) 2^11  (asm statements: hipcc's SLP vectoriser would pair the subtractions into v_pk_add_f32)
    auto res_lo = [](float x, unsigned piece) -> float {
        float f, r, l;
        asm("v_cvt_f32_f16 %0, %1" : "=v"(f) : "v"(piece));
        asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(f));
        asm("v_ldexp_f32 %0, %1, 11" : "=v"(l) : "v"(r));
        return l;
    };
    auto res_hi = [](float x, unsigned piece) -> float {
        float f, r, l;
        asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(piece));
        asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(f));
        asm("v_ldexp_f32 %0, %1, 11" : "=v"(l) : "v"(r));
        return l;
    };
    static_assert(kBxLift == 11 && kBxPx == 2, "the asm statements above carry the lift");
    p[0] = cvt(a, b);
    p[1] = cvt(res_lo(a, p[0]), res_hi(b, p[0]));
}

// The same pieces from the UNSCALED samples (xa, xb) and the window's scale s = 2^k (round 6): the residual (x s - piece) 2^11 = (x s) 2^11 - 2048 piece is an ldexp and ONE
// mixed-precision multiply-add that reads the fp16 piece where it lies (v_fma_mix_f32; exact -- the difference of a float32 and its own nearest fp16 has 14 significant bits,
// and the powers of two only move exponents), where the form above spends a conversion, a subtraction and an ldexp: 8 instructions per pair of samples for 10.
__device__ __forceinline__ void bx_split2s(float xa, float xb, float s, float neg2048, unsigned (&p)[kBxPx])
{
    float sa, sb, ta, tb, ra, rb;
    asm("v_mul_f32 %0, %1, %2" : "=v"(sa) : "v"(xa), "v"(s));
    asm("v_mul_f32 %0, %1, %2" : "=v"(sb) : "v"(xb), "v"(s));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p[0]) : "v"(sa), "v"(sb));
    asm("v_ldexp_f32 %0, %1, 11" : "=v"(ta) : "v"(sa));
    asm("v_ldexp_f32 %0, %1, 11" : "=v"(tb) : "v"(sb));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p[0]), "v"(neg2048), "v"(ta));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p[0]), "v"(neg2048), "v"(tb));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p[1]) : "v"(ra), "v"(rb));
    static_assert(kBxLift == 11 && kBxPx == 2, "the constants above carry the lift");
}

// max (MAX) / min of an unsigned value over the wave, valid in lane 63: four DPP steps inside the rows of 16 lanes (quad permutes, half-row and
// row mirrors), then row_bcast15 into rows 1 and 3 and row_bcast31 into rows 2 and 3
template <bool MAX> __device__ __forceinline__ unsigned bx_wave_reduce(unsigned v)
{
    auto op = [](unsigned a, unsigned b) -> unsigned { return MAX ? max(a, b) : min(a, b); };
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));   // row_mirror
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast15 -> rows 1, 3
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast31 -> rows 2, 3
    return v;
}

__device__ __forceinline__ v4f_bx bx_mfma(uint4 a, uint4 b, v4f_bx c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h_bx, a), __builtin_bit_cast(v8h_bx, b), c, 0, 0, 0);
}

// A window this kernel cannot compute: one that holds an inf / nan (0 x inf = nan on the zero-padded part of the lag blocks, and the
// window's scale is taken from its largest magnitude), or one whose samples span more than 2^24 in magnitude (the fp16 pieces keep 22 bits
// of a sample only down to 2^-29 of the window's largest: a 1e12 glitch would leave its unit-level neighbours 5e-4 of THEIR level).  Such a
// window is noted (careful.hpp) and its outputs -- m in [RS NS w, RS NS (w + 1)) -- are recomputed behind the loop by the reference's own
// float64 sum.

// Persistent 256-thread workgroups: window w+1 is requested into registers before window w is multiplied, so
// its HBM latency hides behind the MFMAs of the same workgroup; the workgroups of a CU run out of phase with
// each other, which is what overlaps the fp16 split / LDS writes / stores of one with the MFMAs of another.
//
// RSP = 2 (row split): the taps and accumulators of a geometry with many row tiles do not fit one wave's registers (L = 12: 96 rows = 6 row
// tiles x 2 blocks: 96 VGPRs of taps + 96 of accumulators).  The waves of a workgroup then pair up: wave w takes the row tiles [RT (w & 1), RT (w & 1) + RT) of the column
// tiles w >> 1, w >> 1 + 2, ... -- RT is the per-wave count, the table holds RT RSP row tiles.  A window fragment is read from the planes
// by both waves of a pair; a wave's 16 RT rows of a column are still one run of y (L = 12: 48 outputs = 384 bytes = three whole lines).
//
// KSP = 4 (lag split; RT = RSP = 1): a decimator with a large M spreads the 16 rows of its one row tile over 16 M input samples, so its lag
// range is 16 M + Ntaps long (M = 12, 512 taps: 22 blocks = 176 VGPRs of taps) while a window holds one or two column tiles -- nothing for
// three of the four waves to do.  The waves then split the LAGS: wave w takes the blocks [KB w, KB w + KB) of every column tile (KB is
// the per-wave count; the table holds 4 KB blocks, the last ones zero-padded), the four partial tiles meet in the LDS, and wave w
// stores column 4 j + w of every lane's four.
// T16 (float32, one row tile of 16 rows with RS = 16: the plain filter): a tile IS 256 consecutive outputs, and the four columns of a lane are 16-output runs
// 64 outputs apart -- a store instruction wrote four 64-byte runs in four different lines (PMC: 298.7 MB written for 268.4).  A 4 x 4 transpose between a lane's four
// registers and the four 16-lane groups (two v_permlane32_swap, two v_permlane16_swap) hands every store instruction 256 consecutive bytes.
template <bool CPLX, int KB, int RT, int RSP, int KSP, bool T16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fir_bx_kernel(const float *__restrict__ x, const uint4 *__restrict__ At, BxArgs a,
                                                     float *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) char bx_smem[];
    __shared__ unsigned long long bx_noted;   // windows for the exact path, by walk step (careful.hpp)
    if (threadIdx.x == 0) bx_noted = 0;       // (the first barrier of the kernel lies between this and any note)
    constexpr int C = CPLX ? 2 : 1;
    static_assert(KSP == 1 || (RT == 1 && RSP == 1), "the lag split serves one-row-tile geometries");
    constexpr int K = 32 * KB * KSP;
    constexpr int UPT = (bx_units(CPLX, KSP > 1) + 255) / 256;   // staged 8-sample units per thread
    constexpr int F4 = 2 * C;             // 16-byte loads per unit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int RTT = RT * RSP;                                   // row tiles of the table
    const int rt0 = RSP > 1 ? RT * __builtin_amdgcn_readfirstlane(wave % RSP) : 0;   // this wave's first
    constexpr int CW = 4 / RSP;                                     // waves that share the column tiles of a window
    const int kb0 = KSP > 1 ? KB * __builtin_amdgcn_readfirstlane(wave) : 0;   // this wave's first 32-lag block
    const int nunits = a.win / 8;
    auto padded = [&](int u) -> int { return a.pad_s ? u + (int)(((unsigned long long)(unsigned)u * a.pad_magic) >> 32) : u; };   // unit -> its place in a plane
    const int plane_bytes = (padded(nunits) + 1) * 16;  // + a dump row for the threads beyond the window

    // Interior windows (16-byte aligned, fully inside [-n_hist, n)) are prefetched into registers one window
    // ahead; the few others (first / last windows, element-aligned views) are staged synchronously by their own
    // code, so that `pre` has a single definition and no register copies (= vmcnt waits) follow the prefetch.
    float4 pre[UPT][F4];
    auto window_g0 = [&](int64_t wdx) { return (int64_t)a.q_ds * a.NS * wdx + a.U0 - (K - 1); };  // input index of element 0
    auto interior = [&](int64_t wdx) {
        const int64_t g0 = window_g0(wdx);
        return (reinterpret_cast<uintptr_t>(x + g0 * C) & 15) == 0 && g0 >= -a.n_hist && g0 + a.win <= a.n;
    };
    // (measured and not kept, round 6: skipping, per wave, a round of 256 units that lies wholly beyond the window -- a window of 840 units is 3.3 rounds -- made
    // every kernel of the family 8 - 15 % SLOWER: the branches around the prefetch registers cost more than the fifth of the staging instructions they save)
    auto load_window = [&](int64_t wdx) {  // interior windows only
        const float *src = x + window_g0(wdx) * C;
#pragma unroll
        for (int h = 0; h < UPT; ++h) {
            const int u = min(tid + 256 * h, nunits - 1);
            const float4 *s4 = reinterpret_cast<const float4 *>(src + (size_t)u * 8 * C);
#pragma unroll
            for (int w = 0; w < F4; ++w) pre[h][w] = s4[w];
        }
    };
    // The window's scale.  fp16 holds 2^-14 .. 2^16, so a window is multiplied by the power of two that puts its largest magnitude
    // into [2^14, 2^15) before it is split (exact), and its outputs by the inverse (times the taps' 2^-te).  wmax_sh[wave]: the waves'
    // maxima of |x| as integer bit patterns (written in front of the barrier that frees the planes, read behind it).
    unsigned *wmax_sh = reinterpret_cast<unsigned *>(bx_smem + (size_t)(kBxPx * C) * plane_bytes + (KSP > 1 ? (size_t)2 * 4 * (4 * C) * 64 * 4 : 0));
    float wscale = 1.f, winv_next = 1.f;
    float neg2048 = -2048.f;
    asm volatile("" : "+v"(neg2048));   // (a register operand: VOP3P takes no literal on this chip)
    bool wbad_next = false;   // the window being staged is one for the exact path (bx_careful_window)
    // Two statistics of the window: its largest magnitude (the scale) and the smallest of the lanes' largest (non-zero) magnitudes -- a window
    // whose samples span more than the fp16 pieces hold goes to the exact path.  Both reductions run on DPP row operations (no LDS round
    // trips: the ds_bpermute chain this replaced was 6 dependent LDS operations per wave with the whole workgroup waiting behind it).
    auto publish_max = [&](unsigned m) {   // m: this thread's maximum of |x| bits
        const unsigned hi = bx_wave_reduce<true>(m), lo = bx_wave_reduce<false>(m ? m : 0xffffffffu);   // (in lane 63)
        if (lane == 63) { wmax_sh[wave] = hi; wmax_sh[4 + wave] = lo; }
    };
    auto fetch_scale = [&]() {   // behind the barrier: wscale for the split, winv_next for this window's outputs
        const unsigned m = max(max(wmax_sh[0], wmax_sh[1]), max(wmax_sh[2], wmax_sh[3]));
        const unsigned lo = min(min(wmax_sh[4], wmax_sh[5]), min(wmax_sh[6], wmax_sh[7]));
        int e = (int)(m >> 23);                       // biased exponent of the largest magnitude
        // not finite, or some lane's samples lie more than 2^24 below the largest: the exact path takes this window's outputs
        wbad_next = e == 255 || (lo != 0xffffffffu && e - (int)(lo >> 23) > 24);
        if (e == 0 || e == 255) e = 141;              // all zero (or not finite): scale 1
        e = e < 16 ? 16 : e;                          // (2^(141 - e) must stay a normal float)
        wscale = __uint_as_float((unsigned)(141 - e + 127) << 23);
        winv_next = __uint_as_float((unsigned)(e - 141 + 127) << 23) * a.tap_inv;
    };
    auto pre_max = [&]() -> unsigned {   // over the prefetched window
        if constexpr (CPLX) {
            // (complex windows keep the bit patterns -- an AND and an integer maximum per scalar: same box, alternating, the float form below made the default .dn by
            // 12 2 % slower, 0.1265 -> 0.1291 ms, while it made the 127-tap float32 filter 1.5 % faster, 0.1124 -> 0.1107)
            unsigned m = 0;
#pragma unroll
            for (int h = 0; h < UPT; ++h)
#pragma unroll
                for (int w = 0; w < F4; ++w) {
                    m = max(m, __float_as_uint(pre[h][w].x) & 0x7fffffffu);
                    m = max(m, __float_as_uint(pre[h][w].y) & 0x7fffffffu);
                    m = max(m, __float_as_uint(pre[h][w].z) & 0x7fffffffu);
                    m = max(m, __float_as_uint(pre[h][w].w) & 0x7fffffffu);
                }
            return m;
        }
        // float32 windows (round 6): |a|, |b| through the source modifiers of ONE v_max3_f32 per pair of samples.  A float maximum drops NaNs, and the window statistic is
        // what notices them: one unordered compare per pair collects them in a scalar mask.
        float mf = 0.f;
        unsigned long long nan_any = 0;
#pragma unroll
        for (int h = 0; h < UPT; ++h)
#pragma unroll
            for (int w = 0; w < F4; ++w) {
                unsigned long long u0, u1;
                asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(mf) : "v"(pre[h][w].x), "v"(pre[h][w].y));
                asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(mf) : "v"(pre[h][w].z), "v"(pre[h][w].w));
                asm("v_cmp_u_f32_e64 %0, %1, %2" : "=s"(u0) : "v"(pre[h][w].x), "v"(pre[h][w].y));
                asm("v_cmp_u_f32_e64 %0, %1, %2" : "=s"(u1) : "v"(pre[h][w].z), "v"(pre[h][w].w));
                nan_any |= u0 | u1;
            }
        unsigned m = __float_as_uint(mf);
        if ((nan_any >> lane) & 1ull) m = 0x7fc00000u;   // (this lane met a NaN: a magnitude with the exponent the scale test looks for)
        return m;
    };
    // 8 samples v[0 .. 8 C) -> one 16-byte row per fp16 piece and component at unit u
    auto split_unit = [&](const float *v, int u) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            unsigned pc[4][kBxPx];
#pragma unroll
            for (int w = 0; w < 4; ++w) bx_split2s(v[(2 * w) * C + c], v[(2 * w + 1) * C + c], wscale, neg2048, pc[w]);
            char *base = bx_smem + (size_t)(kBxPx * c) * plane_bytes + (size_t)padded(u) * 16;
#pragma unroll
            for (int i = 0; i < kBxPx; ++i) *reinterpret_cast<uint4 *>(base + (size_t)i * plane_bytes) = make_uint4(pc[0][i], pc[1][i], pc[2][i], pc[3][i]);
        }
    };
    auto store_window = [&]() {  // the prefetched window (branch-free: units beyond the window go to the dump row)
#pragma unroll
        for (int h = 0; h < UPT; ++h) split_unit(reinterpret_cast<const float *>(&pre[h][0]), min(tid + 256 * h, nunits));
    };
    auto slow_max = [&](int64_t wdx) -> unsigned {   // the same window's largest magnitude
        const int64_t g0 = window_g0(wdx);
        unsigned m = 0;
#pragma unroll 1
        for (int u = tid; u < nunits; u += 256)
#pragma unroll 1
            for (int e = 0; e < 8; ++e) {
                const int64_t g = g0 + 8 * (int64_t)u + e;
                if (g >= -a.n_hist && g < a.n)
#pragma unroll
                    for (int c = 0; c < C; ++c) m = max(m, __float_as_uint(x[g * C + c]) & 0x7fffffffu);
            }
        return m;
    };
    auto stage_window_slow = [&](int64_t wdx) {  // guarded scalar loads, zero outside [-n_hist, n)
        const int64_t g0 = window_g0(wdx);
#pragma unroll 1
        for (int u = tid; u < nunits; u += 256) {
            float v[8 * C];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t g = g0 + 8 * (int64_t)u + e;
                const bool ok = g >= -a.n_hist && g < a.n;
#pragma unroll
                for (int c = 0; c < C; ++c) v[e * C + c] = ok ? x[g * C + c] : 0.f;
            }
            split_unit(v, u);
        }
    };

    const int64_t ncols = (a.n_out + a.RS - 1) / a.RS;
    const int64_t nwin = (ncols + a.NS - 1) / a.NS;
    // XCD-aware walk (as ols_tile_kernel's): workgroup b runs on XCD b % 8, so every XCD takes a contiguous run of the round's windows -- neighbouring
    // windows share their lag range (32 KB - q DS samples: 576 of 6720 for the default .dn by 12 of a 512-tap filter), which then hits that XCD's L2
    // instead of being fetched from HBM by two XCDs (round 5 PMC: 1.09 x the algorithmic bytes for that row, 1.07 x for the 127-tap filter)
    const int64_t w0 = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
    int64_t wdx = w0;
    if (wdx >= nwin) return;
    bool fast = interior(wdx);
    if (fast) load_window(wdx);  // first: the A operands below queue behind it

    // A operands of this lane: [32-lag block][row tile][fp16 piece]
    uint4 areg[KB][RT][kBxPh];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < kBxPh; ++p) areg[kb][rt][p] = At[(((kb0 + kb) * RTT + rt0 + rt) * kBxPh + p) * 64 + lane];

    const int ncol = lane & 15, j = lane >> 4;
    const int ntiles = a.NS / 16;
    // The A operands are settled here: inside the loop the compiler cannot count the loads queued behind them and
    // would drain the prefetch of every iteration in front of the first MFMA.
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < kBxPh; ++p)
                asm volatile("" ::"v"(areg[kb][rt][p].x), "v"(areg[kb][rt][p].y), "v"(areg[kb][rt][p].z), "v"(areg[kb][rt][p].w) : "memory");
    // Iteration w:  MFMAs of window w (stores of all tiles but the wave's last) | barrier | split window w+1
    // into the planes | stores of the last tile | barrier | request window w+2.  The split is the only place
    // that waits on loads, and the only stores still in flight there are a whole tile old (vmcnt retires in
    // order: a wait behind fresh stores would also wait for their acknowledgement).
    publish_max(fast ? pre_max() : slow_max(wdx));
    __syncthreads();
    fetch_scale();
    float winv = winv_next;   // of the window in the planes
    bool wbad = wbad_next;
    if (fast) store_window();
    else stage_window_slow(wdx);
    __syncthreads();
    int64_t wnext = wdx + gridDim.x;
    fast = wnext < nwin && interior(wnext);
    if (fast) load_window(wnext);
#pragma unroll 1
    for (; wdx < nwin;) {
        const int64_t S0 = wdx * a.NS;  // first column of this window
        v4f_bx big[RT][C], small[RT][C];
        // column (within the window) of tile ct, tile row r
        auto col_of = [&](int ct, int r) -> int {
            if (!a.eo) return ct * 16 + r;
            const int mu = r < 4 ? r : (r >= 12 ? r - 8 : r + 4);
            return 32 * (ct >> 1) + 2 * mu + (ct & 1);
        };
        auto mma_tile = [&](int ct) __attribute__((always_inline)) {
            // window element of (column n, block kb, group j, i): q_ds n + 32 kb + 8 j + i
            // (padded planes: unit s col + r lies at s col + r + col + r / s, and r / s = (4 (kb0 + kb)) / s for every lane since 4 divides s)
            const int colr = col_of(ct, ncol);
            const char *bbase = bx_smem + ((size_t)a.q_ds * colr + 8 * j + 32 * kb0) * 2 + (a.pad_s ? 16 * colr : 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int c = 0; c < C; ++c) big[rt][c] = small[rt][c] = v4f_bx{0.f, 0.f, 0.f, 0.f};
            // B operands are re-read for the next block right after their last use in this one (piece 1 is
            // multiplied first, piece 3 last), so every LDS read has >= 3 products (192 cycles) of cover.
            // Small products (relative size 2^-9 .. 2^-18) go to their own accumulator; consecutive MFMAs
            // go to different accumulators (row tile x component).
            uint4 b[C][kBxPx];
            // (pad units in front of block kb0 + kb: wave-uniform)
            auto kpad = [&](int kb) -> int { return __builtin_amdgcn_readfirstlane((int)(((unsigned long long)(unsigned)(4 * (kb0 + kb)) * a.pad_magic) >> 32)); };
            auto read_b = [&](int kb, int p) {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    b[c][p] = *reinterpret_cast<const uint4 *>(bbase + (size_t)(kBxPx * c + p) * plane_bytes + 64 * kb + 16 * kpad(kb));
            };
#pragma unroll
            for (int p = 0; p < kBxPx; ++p) read_b(0, p);
#define SK_BX(PA, PB, ACC)                                                                              \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) _Pragma("unroll") for (int c = 0; c < C; ++c)    \
        ACC[rt][c] = bx_mfma(b[c][PB], areg[kb][rt][PA], ACC[rt][c]);
            // products: (tap piece, signal piece) = (1, 1) into `big`; (2, 1), (1, 2) -- 2^-11 of it, both lifted by 2^11 -- into `small`;
            // (2, 2) is below 2^-22 of the leading product and is not formed
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                SK_BX(1, 0, small)
                SK_BX(0, 0, big)
                __builtin_amdgcn_sched_barrier(0);
                if (kb + 1 < KB) read_b(kb + 1, 0);
                SK_BX(0, 1, small)
                __builtin_amdgcn_sched_barrier(0);
                if (kb + 1 < KB) read_b(kb + 1, 1);
            }
#undef SK_BX
        };
        auto store_tile = [&](int ct) __attribute__((always_inline)) {
            // The window fragment is the A operand and the taps are B, so the tile comes out transposed: lane
            // (r = lane & 15, j) holds row 16 rt + r of the four columns 4 j + i -- the 16 lanes of a group write
            // 16 consecutive outputs (128 bytes of complex64) per store instead of 16-byte pieces 32 bytes apart.
            // (the four columns of a lane are consecutive tile rows 4 j + i: consecutive columns, or every second one)
            const int cstep = a.eo ? 2 * a.RS : a.RS;
            const int64_t m_base = (int64_t)a.RS * (S0 + col_of(ct, 4 * j)) + ncol + 16 * rt0;   // output of (this wave's row tile 0, i = 0)
            float *yb = y + m_base * C;
            const int64_t left = a.n_out - m_base;
            const int rem = left > (int64_t)0x7fffffff ? 0x7fffffff : (left < 0 ? 0 : (int)left);
            // big winv + small (winv 2^-11)  (scalar instructions through asm: hipcc pairs them into v_pk_*_f32, which waits for the matrix pipe)
            const float winv_s = winv * (1.0f / (float)(1 << kBxLift));
            auto sum = [&](float u, float w) -> float {
                float m, r;
                asm("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(u), "v"(winv));
                asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "v"(winv_s), "v"(m));
                return r;
            };
            auto put = [&](int rt, int i, int off) __attribute__((always_inline)) {
                if (CPLX) {
                    v2f_bx o = {sum(big[rt][0][i], small[rt][0][i]), sum(big[rt][C - 1][i], small[rt][C - 1][i])};
                    __builtin_nontemporal_store(o, reinterpret_cast<v2f_bx *>(yb + 2 * off));
                } else {
                    // (float32 outputs leave as 64-byte runs, two store instructions per 128-byte line, and PMC shows 298.7 MB written for 268.4.  Ordinary stores
                    // let the two halves meet in the L2 -- 268.4 MB, traffic 1.001 x -- and are SLOWER: 0.125 against 0.113 ms, round 6; a run-time switch between
                    // the two forms cost another 14 %.  Nontemporal it stays.)
                    __builtin_nontemporal_store(sum(big[rt][0][i], small[rt][0][i]), yb + off);
                }
            };
            // whole tile inside the output and all 16 RT rows in use (uniform): no per-store guards
            if ((a.RS & 15) == 0 && (int64_t)a.RS * (S0 + (a.eo ? 32 * (ct >> 1) + 32 : ct * 16 + 16)) <= a.n_out) {
                if constexpr (T16) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = sum(big[0][0][i], small[0][0][i]);
                    auto swap32 = [](float &e0, float &e1) {   // (register, lane half) transposed
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
                        e0 = __uint_as_float(r[0]);
                        e1 = __uint_as_float(r[1]);
                    };
                    auto swap16 = [](float &e0, float &e1) {   // (register, odd / even row of 16 lanes) transposed
                        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
                        e0 = __uint_as_float(r[0]);
                        e1 = __uint_as_float(r[1]);
                    };
                    swap32(o[0], o[2]);
                    swap32(o[1], o[3]);
                    swap16(o[0], o[1]);
                    swap16(o[2], o[3]);
                    float *yt = y + 16 * (S0 + 16 * ct) + lane;   // register k of lane l: output 64 k + l of the tile
#pragma unroll
                    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(o[k], yt + 64 * k);
                } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) put(rt, i, i * cstep + 16 * rt);
                }
            } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int off = i * cstep + 16 * rt;
                        if (16 * (rt0 + rt) + ncol < a.RS && off < rem) put(rt, i, off);
                    }
            }
        };
        if constexpr (KSP > 1) {
            // every wave multiplies its lags of every column tile; partial tiles [parity][wave][component, i][lane] behind the planes.
            // Column tiles go in PAIRS (round 6): both partial tiles of a pair are written before the one barrier that lets the waves sum them -- and for the
            // window's last pair that barrier is also the one that frees the planes, so the sums of the last pair are formed and stored while window w + 1
            // is being split into the planes.  A window of two column tiles (the default .dn by 12 of a 512-tap filter) passes 2 barriers where it passed 4.
            float *red = reinterpret_cast<float *>(bx_smem + (size_t)(kBxPx * C) * plane_bytes);
            const int cstep = a.eo ? 2 * a.RS : a.RS;
            auto put_partial = [&](int ct) __attribute__((always_inline)) {
                float *mine = red + ((size_t)((ct & 1) * 4 + wave) * (4 * C)) * 64 + lane;
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mine[(c * 4 + i) * 64] = fmaf(small[0][c][i], 1.0f / (float)(1 << kBxLift), big[0][c][i]);
            };
            auto sum_store = [&](int ct, float wi) __attribute__((always_inline)) {
                const float *all = red + ((size_t)((ct & 1) * 4) * (4 * C)) * 64 + lane;
                float o[C];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    o[c] = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) o[c] += all[((size_t)w * (4 * C) + c * 4 + wave) * 64];
                }
                const int64_t m = (int64_t)a.RS * (S0 + col_of(ct, 4 * j)) + ncol + (int64_t)wave * cstep;
                if (ncol < a.RS && m < a.n_out) {
                    if (CPLX) __builtin_nontemporal_store(v2f_bx{o[0] * wi, o[C - 1] * wi}, reinterpret_cast<v2f_bx *>(y + 2 * m));
                    else __builtin_nontemporal_store(o[0] * wi, y + m);
                }
            };
            int ct = 0;
#pragma unroll 1
            for (; ct + 2 < ntiles; ct += 2) {   // all pairs but the last
                mma_tile(ct);
                put_partial(ct);
                mma_tile(ct + 1);
                put_partial(ct + 1);
                __syncthreads();
                sum_store(ct, winv);
                sum_store(ct + 1, winv);
                __syncthreads();                 // (the partial tiles are free again)
            }
            const int nlast = ntiles - ct;       // 1 or 2 tiles
            mma_tile(ct);
            put_partial(ct);
            if (nlast > 1) {
                mma_tile(ct + 1);
                put_partial(ct + 1);
            }
            if (wnext < nwin) publish_max(fast ? pre_max() : slow_max(wnext));
            __syncthreads();  // everyone is done reading the planes, and the last partial tiles are complete
            const float winv_w = winv;   // (still window w's inverse scale: fetch_scale below sets the next one's)
            if (wnext < nwin) {
                fetch_scale();
                if (fast) store_window();
                else stage_window_slow(wnext);
            }
            sum_store(ct, winv_w);
            if (nlast > 1) sum_store(ct + 1, winv_w);
            winv = winv_next;
            __syncthreads();  // the planes hold window w+1 (and the partial tiles are free)
            if (__builtin_expect(wbad, 0)) careful_note(&bx_noted, (wdx - w0) / gridDim.x);
            wbad = wbad_next;
            const int64_t wnext2 = wnext + gridDim.x;
            fast = wnext2 < nwin && interior(wnext2);
            if (fast) load_window(wnext2);
            wdx = wnext;
            wnext = wnext2;
            continue;
        }
        int ct = wave / RSP;
#pragma unroll 1
        for (; ct + CW < ntiles; ct += CW) {
            mma_tile(ct);
            __builtin_amdgcn_s_setprio(3);
            store_tile(ct);
            __builtin_amdgcn_s_setprio(0);
        }
        const bool has_last = ct < ntiles;
        if (has_last) mma_tile(ct);
        if (wnext < nwin) publish_max(fast ? pre_max() : slow_max(wnext));   // (the place that waits for the prefetched window)
        __syncthreads();  // everyone is done reading the planes
        if (wnext < nwin) {
            fetch_scale();
            if (fast) {
                store_window();
            } else {
                stage_window_slow(wnext);
            }
        }
        __builtin_amdgcn_s_setprio(3);
        if (has_last) store_tile(ct);   // (still window w: its own inverse scale)
        __builtin_amdgcn_s_setprio(0);
        winv = winv_next;
        __syncthreads();  // the planes hold window w+1
        if (__builtin_expect(wbad, 0)) careful_note(&bx_noted, (wdx - w0) / gridDim.x);
        wbad = wbad_next;
        const int64_t wnext2 = wnext + gridDim.x;
        fast = wnext2 < nwin && interior(wnext2);
        __builtin_amdgcn_s_setprio(3);
        if (fast) load_window(wnext2);
        __builtin_amdgcn_s_setprio(0);
        wdx = wnext;
        wnext = wnext2;
    }
    const unsigned long long noted = careful_noted(&bx_noted);
    if (__builtin_expect(noted != 0, 0)) {
        int64_t k = 0;
        for (int64_t w = w0; w < nwin; w += gridDim.x, ++k)
            if (careful_step_noted(noted, k))
                careful_fir_range<float, CPLX>(x, y, a.n_hist, a.n_out, (int64_t)a.RS * a.NS * w, (int64_t)a.RS * a.NS, a.L, a.M, a.cf, tid);
    }
}

// ---- host side ------------------------------------------------------------------------------------
// nearest fp16 (ties to even) of a value inside the fp16 range, and back
static unsigned short bx_f16_rne(double v)
{
    const _Float16 hv = (_Float16)v;
    unsigned short u;
    std::memcpy(&u, &hv, 2);
    return u;
}
static double bx_f16_val(unsigned short h)
{
    _Float16 hv;
    std::memcpy(&hv, &h, 2);
    return (double)hv;
}

// does a wave's share fit its 256 VGPRs (2 waves per SIMD)?  A operands (4 per tap piece, 32-lag block and row tile) + accumulators + B
// fragments + 32 prefetch registers + ~60 others; kb, rt: blocks / row tiles PER WAVE.  Used by the geometry below and by the dispatch, so
// that only kernels the geometry can pick are instantiated.
static constexpr bool bx_fits(bool cplx, int kb, int rt)
{
    return kb * rt <= 16 && 4 * kBxPh * kb * rt + 8 * (cplx ? 2 : 1) * rt + 4 * kBxPx * (cplx ? 2 : 1) + 32 + 60 <= 252;
}

// geometry of one (L, M): false if the kernel family does not cover it
static bool bx_geometry(const FirHandle *h, int L, int M, FirHandle::BxTab *t)
{
    const int g = std::gcd(L, M), Lp = L / g, q = M / g;
    const int P = h->ntaps, T = (P + L - 1) / L;
    const int ds0 = 8 / std::gcd(q, 8);  // q DS must be a multiple of 8
    int best_k = 0;
    double best_util = 0.0;
    for (int k = 1; k <= 16; ++k) {
        const int RS = Lp * ds0 * k, RT = (RS + 15) / 16;
        if (RT > 8) break;
        const double util = (double)RS / (16.0 * RT);
        if (util > best_util + 1e-9) { best_util = util; best_k = k; }
    }
    if (best_k == 0 || best_util < 0.74) return false;
    const int comp = dtype_complex(h->dtype) ? 2 : 1;
    const int al = comp == 2 ? 2 : 4;
    const int cap = 8 * (comp == 2 ? kBxUnitsC : kBxUnitsR);
    int DS, RS, RT, U0, KB;
    for (;; best_k /= 2) {
        DS = ds0 * best_k; RS = Lp * DS; RT = (RS + 15) / 16;
        int imax = 0;
        for (int c = 0; c < Lp; ++c) imax = std::max(imax, (int)(((int64_t)c * M) / L));
        U0 = imax + q * (DS - 1);
        // window element 0 is input q_ds S0 + U0 + 1 - 32 KB: a 16-byte boundary of x for U0 + 1 = 0 mod 4 (2 for complex)
        U0 += (al - (U0 + 1) % al) % al;
        KB = (T + U0 + 31) / 32;
        // A decimator with a large M (one class, one row tile): 16 slots per column are 16 M inputs, and 16 columns of them may not fit the
        // window (M = 24: 16 x 384 samples).  Fewer slots per column then -- half-empty row tiles cost matrix-pipe time these shapes do
        // not lack (M = 24, 512 taps, complex64: 1.43 ms per 2^26 inputs on the kernels behind this one).
        if (Lp > 1 || best_k % 2 || RS < 8 || 8 * bx_units(comp == 2, true) >= q * DS * 15 + 32 * ((KB + 3) / 4 * 4)) break;
    }
    // What does not fit one wave (bx_fits) is tried with the row tiles dealt to wave pairs (RSP = 2: see the kernel); four or more row tiles
    // always are (the same speed where both fit -- L = 8, 48 taps per phase: 0.1245 / 0.1259 ms -- and the one-wave forms of 4 x 2, 6 x 1 spilled)
    int RSP = 0;
    const bool pairs_first = RT >= 4 && RT % 2 == 0;
    for (int i = 0; i < 2 && !RSP; ++i) {
        const int rsp = (i == 0) == pairs_first ? 2 : 1;
        if (RT % rsp || (rsp > 1 && RT < 4)) continue;
        if (bx_fits(comp == 2, KB, RT / rsp)) RSP = rsp;
    }
    // One row tile and a window that holds fewer column tiles than the workgroup has waves (a decimator with a large M), or more blocks
    // than one wave's registers take: the waves split the lags (KSP = 4: see the kernel); the table is padded to 4 equal shares.
    int KSP = 1, KBT = KB;
    if (RT == 1) {
        const int ns_max = cap > 32 * KB ? (cap - 32 * KB) / (q * DS) + 1 : 0;
        if (!RSP || ns_max < 64) {
            const int kbw = (KB + 3) / 4;
            if (kbw <= 12 && 8 * bx_units(comp == 2, true) >= 32 * 4 * kbw + q * DS * 15) { KSP = 4; KBT = 4 * kbw; RSP = 1; }
        }
    }
    if (!RSP) return false;
    t->RSP = RSP;
    t->KSP = KSP;
    t->L = L; t->M = M; t->Lp = Lp; t->q = q; t->DS = DS; t->RS = RS; t->RT = RT; t->U0 = U0; t->KB = KBT; t->At = nullptr;
    return true;
}

static int bx_columns(const FirHandle::BxTab *t, int comp, int64_t n_out)
{
    // columns per workgroup: what a window of kBxUnitsC / kBxUnitsR 8-sample units holds (32 KiB of fp16 planes: 3
    // workgroups per CU by LDS, 2 by registers), multiples of 64 (16 for wide strides), at most 512
    auto win_of = [&](int NS) { return ((t->q * t->DS * (NS - 1) + 32 * t->KB) + 7) / 8 * 8; };
    const int cap = 8 * bx_units(comp == 2, t->KSP > 1);
    int NS = 512;
    while (NS > 64 && win_of(NS) > cap) NS -= 64;
    while (NS > 16 && win_of(NS) > cap) NS -= 16;
    if (win_of(NS) > cap) return 0;
    const int64_t ncols = (n_out + t->RS - 1) / t->RS;
    while (NS > 64 && (ncols + NS - 1) / NS < 2 * ctx().num_cus) NS -= 64;  // small problems: more windows
    return NS;
}

bool fir_bx_supported(const FirHandle *h, int L, int M, int64_t n_out)
{
    if (h->taps_complex || dtype_double(h->dtype)) return false;
    FirHandle::BxTab t;
    if (!bx_geometry(h, L, M, &t)) return false;
    if (bx_columns(&t, dtype_complex(h->dtype) ? 2 : 1, n_out) == 0) return false;
    return n_out >= (int64_t)t.RS * 64;
}

// 32-lag blocks the kernel would run for (L, M); 0: not covered (cost model of the callers)
int fir_bx_blocks(const FirHandle *h, int L, int M, int *row_tiles)
{
    if (h->taps_complex || dtype_double(h->dtype)) return 0;
    if (!opt().fir_bx || !opt().fir_mm) return 0;
    FirHandle::BxTab t;
    if (!bx_geometry(h, L, M, &t)) return 0;
    if (row_tiles) *row_tiles = t.RT;
    return t.KB;
}

// A-operand table of one (L, M): At[((kb RT + rt) kBxPh + piece) 64 + lane] = 8 fp16 of row 16 rt + (lane & 15),
// lags u = K - 1 - (32 kb + 8 (lane >> 4) + i); the taps are L b 2^te, te the power of two that puts the largest into [2^13, 2^14)
static int get_bx_table(FirHandle *h, int L, int M, const FirHandle::BxTab **out)
{
    for (auto &t : h->bx)
        if (t.L == L && t.M == M) { *out = &t; return SKDSP_OK; }
    FirHandle::BxTab t;
    SK_CHECK(bx_geometry(h, L, M, &t), SKDSP_ERR_UNSUPPORTED, "fir_bx: L=%d M=%d not covered", L, M);
    const int P = h->ntaps, T = (P + L - 1) / L, K = 32 * t.KB;
    double tmax = 0.0;
    for (int k = 0; k < P; ++k) tmax = std::max(tmax, std::fabs((double)L * h->taps_host[k]));
    int te = 0;
    if (tmax > 0.0 && std::isfinite(tmax)) {
        int ex;
        std::frexp(tmax, &ex);        // tmax = m 2^ex, m in [0.5, 1)
        te = std::min(std::max(14 - ex, -100), 100);
    }
    t.tap_inv = (float)std::ldexp(1.0, -te);
    std::vector<unsigned short> host((size_t)t.KB * t.RT * kBxPh * 64 * 8, 0);
    for (int kb = 0; kb < t.KB; ++kb)
        for (int rt = 0; rt < t.RT; ++rt)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int r = 16 * rt + (lane & 15);
                    if (r >= t.RS) continue;
                    const int u = K - 1 - (32 * kb + 8 * (lane >> 4) + i);
                    const int ds = r / t.Lp, c = r % t.Lp;
                    const int64_t cm = (int64_t)c * M;
                    const int phi = (int)(cm % L), ic = (int)(cm / L);
                    const int tt = u - t.U0 + ic + t.q * ds;
                    if (tt < 0 || tt >= T) continue;
                    const int k = phi + L * tt;
                    if (k >= P) continue;
                    double v = std::ldexp((double)L * h->taps_host[k], te);
                    for (int p = 0; p < kBxPh; ++p) {   // (the second piece: the residual lifted by 2^11, see kBxLift)
                        const unsigned short piece = bx_f16_rne(v);
                        host[((((size_t)kb * t.RT + rt) * kBxPh + p) * 64 + lane) * 8 + i] = piece;
                        v = std::ldexp(v - bx_f16_val(piece), kBxLift);
                    }
                }
    SK_HIP(hipMalloc(&t.At, host.size() * 2));
    SK_HIP(hipMemcpy(t.At, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    h->bx.push_back(t);
    *out = &h->bx.back();
    return SKDSP_OK;
}

template <bool CPLX, int KB, int RT, int RSP, int KSP = 1, bool T16 = false>
static void bx_launch_one(unsigned grid, size_t lds, hipStream_t s, const void *x, const void *At, const BxArgs &a, void *y)
{
    if (lds > (size_t)64 * 1024)
        (void)hipFuncSetAttribute((const void *)fir_bx_kernel<CPLX, KB, RT, RSP, KSP, T16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((fir_bx_kernel<CPLX, KB, RT, RSP, KSP, T16>), dim3(grid), dim3(256), lds, s, (const float *)x, (const uint4 *)At, a, (float *)y);
}

// (KB: 32-lag blocks per wave, RT: row tiles per wave)
template <bool CPLX, int KB, int RT, int RSP, int KSP>
static bool bx_launch_if(unsigned grid, size_t lds, hipStream_t s, const void *x, const void *At, const BxArgs &a, void *y)
{
    if constexpr (bx_fits(CPLX, KB, RT)) {
        // (float32, one row tile, RS = 16 -- the plain filter up to ~240 taps: the tile is a run of 256 outputs, stored as such; see T16 at the kernel)
        if constexpr (!CPLX && RT == 1 && RSP == 1 && KSP == 1 && KB <= 8) {
            if (a.RS == 16 && !a.eo && opt().fir_bx_t16) {
                bx_launch_one<CPLX, KB, RT, RSP, KSP, true>(grid, lds, s, x, At, a, y);
                return true;
            }
        }
        bx_launch_one<CPLX, KB, RT, RSP, KSP>(grid, lds, s, x, At, a, y);
        return true;
    } else {
        return false;
    }
}
template <bool CPLX>
static bool bx_dispatch(int KB, int RT, int RSP, int KSP, unsigned grid, size_t lds, hipStream_t s, const void *x, const void *At, const BxArgs &a, void *y)
{
#define SK_BXC(kb, rt) case (kb) * 16 + (rt): return bx_launch_if<CPLX, kb, rt, 1, 1>(grid, lds, s, x, At, a, y);
#define SK_BXR(kb, rt) case (kb) * 16 + (rt): return bx_launch_if<CPLX, kb, rt, 2, 1>(grid, lds, s, x, At, a, y);
#define SK_BXK(kb) case (kb): return bx_launch_if<CPLX, kb, 1, 1, 4>(grid, lds, s, x, At, a, y);
    if (KSP == 4) {
        switch (KB) {
            SK_BXK(1) SK_BXK(2) SK_BXK(3) SK_BXK(4) SK_BXK(5) SK_BXK(6) SK_BXK(7) SK_BXK(8) SK_BXK(9) SK_BXK(10) SK_BXK(11) SK_BXK(12)
        default: return false;
        }
    }
    // (the lists are what bx_fits can accept for either dtype; bx_launch_if instantiates only what fits)
    if (RSP == 1) {   // one, two, three, five or seven row tiles
        switch (KB * 16 + RT) {
            SK_BXC(1, 1) SK_BXC(2, 1) SK_BXC(3, 1) SK_BXC(4, 1) SK_BXC(5, 1) SK_BXC(6, 1) SK_BXC(7, 1) SK_BXC(8, 1) SK_BXC(9, 1) SK_BXC(10, 1) SK_BXC(11, 1) SK_BXC(12, 1)
            SK_BXC(13, 1) SK_BXC(14, 1) SK_BXC(15, 1) SK_BXC(16, 1)
            SK_BXC(1, 2) SK_BXC(2, 2) SK_BXC(3, 2) SK_BXC(4, 2) SK_BXC(5, 2) SK_BXC(6, 2) SK_BXC(7, 2) SK_BXC(8, 2)
            SK_BXC(1, 3) SK_BXC(2, 3) SK_BXC(3, 3) SK_BXC(4, 3) SK_BXC(5, 3)
            SK_BXC(1, 5) SK_BXC(2, 5)
            SK_BXC(1, 7)
        default: return false;
        }
    }
    switch (KB * 16 + RT) {   // four, six or eight row tiles: half of them per wave
        SK_BXR(1, 2) SK_BXR(2, 2) SK_BXR(3, 2) SK_BXR(4, 2) SK_BXR(5, 2) SK_BXR(6, 2) SK_BXR(7, 2) SK_BXR(8, 2)
        SK_BXR(1, 3) SK_BXR(2, 3) SK_BXR(3, 3) SK_BXR(4, 3) SK_BXR(5, 3)
        SK_BXR(1, 4) SK_BXR(2, 4) SK_BXR(3, 4)
    default: return false;
    }
#undef SK_BXC
#undef SK_BXR
#undef SK_BXK
}

int fir_bx_launch(FirHandle *h, const void *x, int64_t n, int64_t n_hist, int L, int M, int64_t n_out, void *y, hipStream_t s)
{
    note_path("fir_bx");
    if (n_out <= 0) return SKDSP_OK;
    const FirHandle::BxTab *t = nullptr;
    int rc = get_bx_table(h, L, M, &t);
    if (rc) return rc;
    const bool cplx = dtype_complex(h->dtype);
    BxArgs a;
    a.n = n; a.n_hist = n_hist; a.n_out = n_out;
    a.q_ds = t->q * t->DS; a.RS = t->RS; a.U0 = t->U0;
    a.NS = bx_columns(t, cplx ? 2 : 1, n_out);
    SK_CHECK(a.NS > 0, SKDSP_ERR_UNSUPPORTED, "fir_bx: window does not fit LDS (L=%d M=%d)", L, M);
    a.win = ((a.q_ds * (a.NS - 1) + 32 * t->KB) + 7) / 8 * 8;
    a.eo = ((a.q_ds / 8) & 1) && a.NS % 32 == 0 ? 1 : 0;
    const int su = a.q_ds / 8;
    a.pad_s = su % 4 == 0 ? su : 0;
    a.pad_magic = a.pad_s ? (unsigned)(((1ull << 32) + su - 1) / su) : 0u;
    const int units = a.win / 8;
    size_t lds = (size_t)(cplx ? 2 : 1) * kBxPx * (size_t)(units + (a.pad_s ? units / su : 0) + 1) * 16;  // (+ a dump row per plane)
    if (t->KSP > 1) lds += (size_t)2 * 4 * (cplx ? 8 : 4) * 64 * sizeof(float);   // the partial tiles of the lag split, two generations
    lds += 32;                                                                      // the waves' window statistics
    a.tap_inv = t->tap_inv;
    a.L = L; a.M = M;
    if ((rc = fir_careful(h, &a.cf))) return rc;
    const int64_t ncols = (n_out + a.RS - 1) / a.RS;
    const int64_t nwin = (ncols + a.NS - 1) / a.NS;
    const unsigned grid = (unsigned)std::min<int64_t>(nwin, (int64_t)2 * ctx().num_cus);  // persistent: two per CU
    const bool ok = cplx ? bx_dispatch<true>(t->KB / t->KSP, t->RT / t->RSP, t->RSP, t->KSP, grid, lds, s, x, t->At, a, y)
                         : bx_dispatch<false>(t->KB / t->KSP, t->RT / t->RSP, t->RSP, t->KSP, grid, lds, s, x, t->At, a, y);
    SK_CHECK(ok, SKDSP_ERR_UNSUPPORTED, "fir_bx: no kernel for %d blocks x %d row tiles", t->KB, t->RT);
    SK_HIP(hipGetLastError());
    return SKDSP_OK;
}

}  // namespace skdsp
