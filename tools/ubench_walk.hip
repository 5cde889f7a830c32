// What does the memory walk of the overlap-save headline cost on its own, and what shapes it?  A persistent kernel with the headline's access pattern and no
// arithmetic: tiles of N = 8192 complex64 inputs (64 KiB, 16 nontemporal 16-byte loads per thread of a 256-thread workgroup, 4 KiB per load instruction), V = 7168
// outputs (14 nontemporal 16-byte stores per thread), the next tile's loads requested before this tile's stores, XCD-contiguous walk, `wgs` workgroups per CU,
// an optional pause in front of the loads and of the stores (the transforms' time: ~20 000 clocks per tile in the real kernel) and optional workgroup barriers (3 per tile like the real kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_walk.hip -o tools/ubench_walk && tools/ubench_walk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int BARRIERS>
__global__ __launch_bounds__(256) void walk(const v4f *__restrict__ x, v4f *__restrict__ y, int64_t ntiles, int pause, int keep)
{
    const int t = threadIdx.x;
    int64_t tile = (gridDim.x % 8 == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
    v4f v[16], nx[16];
    auto load = [&](int64_t tl, v4f *d) {
        const v4f *src = x + tl * 3584;   // V = 7168 samples = 3584 float4
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            if (a < keep || a >= 16 - keep) d[a] = src[a * 256 + t];
            else d[a] = __builtin_nontemporal_load(src + a * 256 + t);
        }
    };
    if (tile < ntiles) load(tile, v);
    for (; tile < ntiles; tile += gridDim.x) {
        if (BARRIERS) __syncthreads();
        for (int i = 0; i < pause; ++i) __builtin_amdgcn_s_sleep(16);
        const int64_t next = tile + gridDim.x;
        if (next < ntiles) load(next, nx);
        if (BARRIERS) __syncthreads();
        for (int i = 0; i < pause; ++i) __builtin_amdgcn_s_sleep(16);
#pragma unroll
        for (int a = 0; a < 16; ++a) asm volatile("" :: "v"(nx[a]));
        v4f *dst = y + tile * 3584;
#pragma unroll
        for (int a = 2; a < 16; ++a) __builtin_nontemporal_store(v[a], dst + (a - 2) * 256 + t);
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = nx[a];
        if (BARRIERS) __syncthreads();
    }
}
int main()
{
    const int64_t n = 1 << 26;                     // complex64 samples
    const int64_t ntiles = (n - 1024) / 7168;
    v4f *x, *y;
    hipMalloc(&x, n * 8 + 65536); hipMalloc(&y, n * 8 + 65536);
    hipMemset(x, 0, n * 8 + 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = 16.0 * 7168 * ntiles;     // algorithmic: 8 B in + 8 B out per output sample
    for (int barriers = 0; barriers < 2; ++barriers)
        for (int wgs : {1, 2, 3, 4})
            for (int pause : {0, 5, 10, 15})
                for (int keep : {0, 2}) {
                    if (keep && (pause != 10 || wgs != 2)) continue;
                    float best = 1e30f;
                    for (int rep = 0; rep < 6; ++rep) {
                        hipEventRecord(e0);
                        if (barriers) hipLaunchKernelGGL(walk<1>, dim3(256 * wgs), dim3(256), 0, 0, x, y, ntiles, pause, keep);
                        else hipLaunchKernelGGL(walk<0>, dim3(256 * wgs), dim3(256), 0, 0, x, y, ntiles, pause, keep);
                        hipEventRecord(e1); hipEventSynchronize(e1);
                        float ms; hipEventElapsedTime(&ms, e0, e1);
                        if (rep > 1 && ms < best) best = ms;
                    }
                    printf("barriers %d  workgroups per CU %d  pause 2 x %2d x 1024 clocks per tile  overlap blocks kept in L2 %d:  %.4f ms  %.2f TB/s algorithmic (%.1f %% of 8)\n",
                           barriers, wgs, pause, keep, best, bytes / best / 1e9, bytes / best / 1e9 / 80);
                }
    return 0;
}
