// v_mfma_f64_4x4x4_4b_f64 on gfx950: where its operands live (one-hot probe) and what it costs next to v_mfma_f64_16x16x4_f64.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_f64_4x4.hip -o /tmp/ub44 && /tmp/ub44
// Why: the parallel-form IIR kernel forms the from-rest end states V = G x on the FP64 matrix pipe; with 4 biquads (8 state rows) half of
// the 16 rows of the 16x16x4 instruction are padding, and the instruction costs 64 cycles whatever its rows hold.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned long long *mask)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) mask[la * 64 + lb] = m;
        }
}
__global__ __launch_bounds__(256) void rate(int which, int iters, double *out)
{
    double a = threadIdx.x * 1e-3, c = 0.999;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    v4d m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    if (which == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, a, s0, 0, 0, 0); s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, a, s1, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, a, s2, 0, 0, 0); s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, a, s3, 0, 0, 0);
                s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, a, s4, 0, 0, 0); s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, a, s5, 0, 0, 0);
                s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, a, s6, 0, 0, 0); s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, a, s7, 0, 0, 0);
            }
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a, m1, 0, 0, 0);
                m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a, m2, 0, 0, 0); m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a, m3, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + m0[0] + m1[1] + m2[2] + m3[3];
}
int main()
{
    unsigned long long *mask, h[4096];
    hipMalloc(&mask, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mask);
    hipMemcpy(h, mask, sizeof(h), hipMemcpyDeviceToHost);
    // for every output lane: the (A lane, B lane) pairs that feed it
    printf("v_mfma_f64_4x4x4_4b: D lane <- sum over (A lane, B lane):\n");
    for (int d = 0; d < 64; ++d) {
        printf("  D[%2d] <-", d);
        for (int la = 0; la < 64; ++la)
            for (int lb = 0; lb < 64; ++lb)
                if (h[la * 64 + lb] >> d & 1) printf(" (%d,%d)", la, lb);
        printf("\n");
    }
    double *out;
    hipMalloc(&out, 256 * 4 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, which, iters, out);   // one wave per SIMD
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%s: %.3f ms for %d instructions per wave = %.1f ns each (one wave per SIMD)\n", which ? "v_mfma_f64_16x16x4" : "v_mfma_f64_4x4x4_4b", ms, 16 * iters, ms * 1e6 / (16.0 * iters));
        }
    return 0;
}
