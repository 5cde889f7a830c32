// ubench_fp64b.hip -- v_fma_f64 throughput by operand kind (SGPR vs VGPR sources), 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: z = fma(z, s, s)   1 VGPR source
// MODE 1: z = fma(s, w, z)   2 VGPR sources (w loop-invariant VGPR)      <- the IIR's v_fmac_f64 shape
// MODE 2: z = fma(u, w, z)   3 VGPR sources
// MODE 3: z = z * s          v_mul_f64
// MODE 4: IIR section shape: y=fma(b0,x,z0); z0=fma(b1,x,fma(-a1,y,z1)); z1=fma(b2,x,-a2*y); x=y  over 8 sections
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters)
{
    constexpr int C = 8;
    double z[C], w[C], u[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { z[c] = threadIdx.x * 1e-3 + c; w[c] = 1.0 + 1e-9 * (threadIdx.x + c); u[c] = 0.999 + 1e-9 * c * threadIdx.x; }
    if (MODE == 4) {
        double z1[C];
#pragma unroll
        for (int c = 0; c < C; ++c) z1[c] = 0.5 * c;
        double x = 1e-3 * threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                double xin = x * 0.5 + uu;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const double y = fma(a, xin, z[c]);
                    z[c] = fma(b, xin, fma(-a, y, z1[c]));
                    z1[c] = fma(a, xin, -b * y);
                    xin = y;
                }
                x = xin;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) z[c] += z1[c];
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (MODE == 0) z[c] = fma(z[c], a, b);
                    if (MODE == 1) z[c] = fma(a, w[c], z[c]);
                    if (MODE == 2) z[c] = fma(u[c], w[c], z[c]);
                    if (MODE == 3) z[c] = z[c] * a;
                }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += z[c] + w[c] + u[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> static void run(double *out, const char *what)
{
    const int iters = 4096, grid = 512;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 0.999999, 1e-7, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)grid * 256 * iters * (MODE == 4 ? 4 * 8 * 5 : 8 * 8);
        if (rep == 2) printf("%-34s %7.3f ms  %6.2f T dp-op/s  (x2 = %.1f TFLOP/s)\n", what, ms, ops / ms / 1e9, 2 * ops / ms / 1e9);
    }
}

int main()
{
    double *out;
    (void)hipMalloc(&out, 512 * 256 * 8);
    run<0>(out, "fma(v, s, s)  1 VGPR src");
    run<1>(out, "fma(s, v, v)  2 VGPR src");
    run<2>(out, "fma(v, v, v)  3 VGPR src");
    run<3>(out, "mul(v, s)");
    run<4>(out, "8-biquad DF2T cascade shape");
    return 0;
}
