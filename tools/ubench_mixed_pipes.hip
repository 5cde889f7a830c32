// Does an FP32-input matrix instruction (v_mfma_f32_16x16x4_f32, v_mfma_f32_16x16x32_f16) of one wave run BESIDE the v_fma_f64 stream of the
// other wave of its SIMD on gfx950 -- unlike v_mfma_f64_16x16x4_f64, which shares the FP64 datapath (tools/ubench_dp_pipes.hip)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mixed_pipes.hip -o /tmp/ubench_mixed_pipes && /tmp/ubench_mixed_pipes
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) run role A, waves 4-7 (the second wave of each SIMD) role B:
// 0 idle, 1 v_fma_f64 stream, 2 f64 MFMA, 3 f32 MFMA 16x16x4, 4 f16 MFMA 16x16x32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void k(int roleA, int roleB, int iters, double *out)
{
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    double a0 = threadIdx.x, a1 = 1.0, a2 = 2.0, a3 = 3.0, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const double c = 0.999999, d = 1e-9;
    v4d m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    v4f f0 = {0, 0, 0, 0}, f1 = f0, f2 = f0, f3 = f0;
    const float cf = 0.999f, bf = (float)threadIdx.x;
    v8h ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.5f + i); hb[i] = (_Float16)(0.25f * threadIdx.x); }
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
            for (int u = 0; u < 4; ++u) {
                m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a0, m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a1, m1, 0, 0, 0);
                m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a2, m2, 0, 0, 0);
                m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, a3, m3, 0, 0, 0);
            }
        }
    } else if (role == 3) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cf, bf, f0, 0, 0, 0);
                f1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cf, bf, f1, 0, 0, 0);
                f2 = __builtin_amdgcn_mfma_f32_16x16x4f32(cf, bf, f2, 0, 0, 0);
                f3 = __builtin_amdgcn_mfma_f32_16x16x4f32(cf, bf, f3, 0, 0, 0);
            }
        }
    } else if (role == 4) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, f0, 0, 0, 0);
                f1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, f1, 0, 0, 0);
                f2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, f2, 0, 0, 0);
                f3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, f3, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0[0] + m1[1] + m2[2] + m3[3] + f0[0] + f1[1] + f2[2] + f3[3];
}
int main()
{
    double *out;
    hipMalloc(&out, 256 * 512 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char *nm[] = {"idle    ", "fma_f64 ", "mfma_f64", "mfma_f32", "mfma_f16"};
    const int modes[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {1, 1}, {1, 2}, {1, 3}, {3, 1}, {1, 4}, {4, 1}, {3, 3}, {4, 4}};
    for (auto &m : modes) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, m[0], m[1], iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        // per iteration: role 1 = 64 FMA wave-instructions, roles 2-4 = 16 matrix instructions
        printf("first wave of a SIMD: %s second: %s ->  %.3f ms  (%.1f clocks per iteration at 2.4 GHz)\n", nm[m[0]], nm[m[1]], best, best * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
