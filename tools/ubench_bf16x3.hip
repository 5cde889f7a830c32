// Precision + rate probe: f32 dot products on the BF16 matrix pipe through a 3-way bf16 split
// (x = x1 + x2 + x3, 8 mantissa bits each; the six products of order <= 2^-16 kept) against the
// FP32 matrix pipe (v_mfma_f32_16x16x4_f32) and a float64 host reference.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_bf16x3 tools/ubench_bf16x3.hip && ./ubench_bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));

__device__ inline unsigned short bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ inline float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ inline void split3(float x, unsigned short &a, unsigned short &b, unsigned short &c)
{
    a = bf16_rne(x);
    float r = x - bf16_f(a);
    b = bf16_rne(r);
    r -= bf16_f(b);
    c = bf16_rne(r);
}

// A: [16][K] row-major, B: [K][16]; one wave per tile; out C [16][16]
template <int MODE>
__global__ void probe(const float *A, const float *B, float *C, int K, int reps)
{
    const int lane = threadIdx.x & 63;
    const size_t tile = blockIdx.x;
    A += tile * 16 * K; B += tile * K * 16; C += tile * 256;
    const int c = lane & 15, j = lane >> 4;
    v4f acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 0) {
            for (int k0 = 0; k0 < K; k0 += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c * K + k0 + j], B[(k0 + j) * 16 + c], acc, 0, 0, 0);
        } else {
            for (int k0 = 0; k0 < K; k0 += 32) {
                v8s a1, a2, a3, b1, b2, b3;
                for (int i = 0; i < 8; ++i) {
                    unsigned short p, q, r;
                    split3(A[c * K + k0 + 8 * j + i], p, q, r);
                    a1[i] = p; a2[i] = q; a3[i] = r;
                    split3(B[(k0 + 8 * j + i) * 16 + c], p, q, r);
                    b1[i] = p; b2[i] = q; b3[i] = r;
                }
#define MM(a, b, acc) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), acc, 0, 0, 0)
                if (MODE == 1) {  // small terms first into their own accumulator
                    MM(a3, b1, acc2); MM(a1, b3, acc2); MM(a2, b2, acc2);
                    MM(a2, b1, acc2); MM(a1, b2, acc2);
                    MM(a1, b1, acc);
                } else if (MODE == 2) {  // everything into one accumulator
                    MM(a3, b1, acc); MM(a1, b3, acc); MM(a2, b2, acc);
                    MM(a2, b1, acc); MM(a1, b2, acc);
                    MM(a1, b1, acc);
                } else {  // 3 products only
                    MM(a2, b1, acc2); MM(a1, b2, acc2);
                    MM(a1, b1, acc);
                }
            }
        }
    }
    for (int r = 0; r < 4; ++r) C[(4 * j + r) * 16 + c] = acc[r] + acc2[r];
}

int main()
{
    const int K = 160, T = 4096;
    std::vector<float> A((size_t)T * 16 * K), B((size_t)T * K * 16);
    srand(7);
    auto rnd = [] { return (rand() + 0.5) / (RAND_MAX + 1.0); };
    auto gauss = [&] { return std::sqrt(-2 * std::log(rnd())) * std::cos(6.283185307179586 * rnd()); };
    for (auto &v : A) v = (float)(gauss() * std::exp(-4.0 * rnd()));
    for (auto &v : B) v = (float)gauss();
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)T * 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref((size_t)T * 256);
    double cmax = 0;
    for (int t = 0; t < T; ++t)
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)A[((size_t)t * 16 + r) * K + k] * (double)B[((size_t)t * K + k) * 16 + c];
                ref[(size_t)t * 256 + r * 16 + c] = s;
                cmax = std::fmax(cmax, std::fabs(s));
            }
    std::vector<float> C((size_t)T * 256);
    const char *names[] = {"f32 16x16x4 (one accumulator)", "bf16x3, 6 products, small terms apart", "bf16x3, 6 products, one accumulator",
                           "bf16x3, 3 products"};
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(dC, 0, (size_t)T * 256 * 4);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K, 1);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K, 1);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K, 1);
        if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K, 1);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, e2 = 0;
        for (size_t i = 0; i < C.size(); ++i) {
            const double e = std::fabs((double)C[i] - ref[i]);
            emax = std::fmax(emax, e);
            e2 += e * e;
        }
        printf("%-44s max err / max|C| = %.3e   rms err / max|C| = %.3e\n", names[mode], emax / cmax, std::sqrt(e2 / C.size()) / cmax);
    }
    // float32 rounding of the exact result, for scale
    double emax = 0;
    for (size_t i = 0; i < ref.size(); ++i) emax = std::fmax(emax, std::fabs((double)(float)ref[i] - ref[i]));
    printf("%-44s max err / max|C| = %.3e\n", "(float32 rounding of the exact result)", emax / cmax);
    return 0;
}
