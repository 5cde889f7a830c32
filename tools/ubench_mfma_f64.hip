// ubench_mfma_f64.hip -- sustained v_mfma_f64_16x16x4_f64 rate (2048 flop per wave-instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int ACC>
__global__ __launch_bounds__(256) void k(double *out, int iters)
{
    v4d c[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) c[i] = v4d{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-6, b = 0.5 + threadIdx.x * 1e-7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ACC> static void run(double *out, int wg_per_cu)
{
    const int iters = 4096, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<ACC>, dim3(grid), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 4 * iters * 4 * ACC * 2048.0;
        if (rep == 2) printf("acc chains %d, waves/SIMD %d: %.3f ms  %.1f TFLOP/s fp64 (matrix)\n", ACC, wg_per_cu, ms, flop / ms / 1e9);
    }
}

int main()
{
    double *out;
    (void)hipMalloc(&out, 2048 * 256 * 8);
    for (int w : {1, 2, 4}) {
        run<1>(out, w);
        run<2>(out, w);
        run<4>(out, w);
    }
    return 0;
}
