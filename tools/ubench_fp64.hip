// ubench_fp64.hip -- what does v_fma_f64 really sustain on this chip, as a function of waves per
// SIMD and of independent chains per thread?  (Bounds the IIR scan kernels: DESIGN.md 4.3.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64.hip -o /tmp/ubench_fp64 && /tmp/ubench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ __launch_bounds__(256) void dfma(double *out, double a, double b, int iters)
{
    double z[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) z[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) z[c] = fma(z[c], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += z[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS> static void run(int wg_per_cu, double *out, int iters = 2048)
{
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(dfma<CHAINS>, dim3(grid), dim3(256), 0, 0, out, 0.999999, 1e-7, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fma_count = (double)grid * 256 * iters * 8 * CHAINS;
        if (rep == 2)
            printf("chains %2d  waves/SIMD %d : %7.3f ms  %6.2f TFLOP/s fp64\n", CHAINS, wg_per_cu, ms, 2 * fma_count / ms / 1e9);
    }
}

int main()
{
    double *out;
    hipMalloc(&out, 256 * 8 * 256 * 8);
    for (int w : {1, 2, 4, 8}) {
        run<1>(w, out);
        run<2>(w, out);
        run<4>(w, out);
        run<8>(w, out);
        run<16>(w, out);
    }
    // sustained: ~100 ms launches (does the clock hold under continuous FP64 FMA?)
    for (int rep = 0; rep < 3; ++rep) run<8>(2, out, 2048 * 256);
    return 0;
}
