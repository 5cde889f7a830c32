// Do v_fma_f64 (vector ALU) and v_mfma_f64_16x16x4_f64 (matrix pipe) of two waves sharing a SIMD run concurrently on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dp_pipes.hip -o /tmp/ubench_dp_pipes && /tmp/ubench_dp_pipes
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) run the role of bit 0, waves 4-7 (the second wave of each
// SIMD) the role of bit 1 of `mode`: 1 = FMA stream, 2 = MFMA stream, 0 = idle.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int roleA, int roleB, int iters, double *out)
{
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    double a0 = threadIdx.x, a1 = 1.0, a2 = 2.0, a3 = 3.0, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const double c = 0.999999, d = 1e-9;
    v4d m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    if (role == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = fma(a2, c, d); a3 = fma(a3, c, d);
                a4 = fma(a4, c, d); a5 = fma(a5, c, d); a6 = fma(a6, c, d); a7 = fma(a7, c, d);
            }
        }
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // 16 MFMAs x 64 cycles = as long as 64 FMAs x 4 cycles x 4
                m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a0, m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a1, m1, 0, 0, 0);
                m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a2, m2, 0, 0, 0);
                m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a3, m3, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0[0] + m1[1] + m2[2] + m3[3];
}
int main()
{
    double *out;
    hipMalloc(&out, 256 * 512 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const int modes[][2] = {{1, 0}, {0, 1}, {2, 0}, {0, 2}, {1, 1}, {2, 2}, {1, 2}, {2, 1}};
    for (auto &m : modes) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, m[0], m[1], iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) {
                // role 1: 64 FMA wave-instructions per iteration; role 2: 16 MFMAs (1024 FMA each)
                printf("first wave of a SIMD: %s, second: %s  ->  %.3f ms", m[0] == 1 ? "FMA " : m[0] == 2 ? "MFMA" : "idle", m[1] == 1 ? "FMA " : m[1] == 2 ? "MFMA" : "idle", ms);
                double fl = 0;
                for (int r = 0; r < 2; ++r) fl += m[r] == 1 ? 2.0 * 64 * 64 * 4 * 256 * iters : m[r] == 2 ? 2.0 * 16 * 1024 * 4 * 256 * iters : 0;
                printf("   %.1f TFLOP/s\n", fl / ms / 1e9);
            }
        }
    }
    return 0;
}
