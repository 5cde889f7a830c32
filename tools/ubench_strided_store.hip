// ubench_strided_store.hip -- what partial-row stores cost on MI355X.  A 512 MiB buffer is written once in total by every variant:
// rows of S bytes, each written as S / 32 pieces of 32 bytes (two 16-byte stores of one lane, or one 16-byte store of each lane of a
// pair); the pieces of a row leave either back to back from the same wave (burst) or piece by piece over the whole buffer (sweep: a
// row's second piece is written long after its first).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_strided_store.hip -o /tmp/ubss && /tmp/ubss
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));

// mode 0: lane = row, writes piece p as two 16-byte stores (the unstaged store of up4k_kernel)
// mode 1: lane pair = row, each lane one 16-byte half of piece p (the staged store)
// sweep = 1: the grid walks all rows for piece 0, then all rows for piece 1, ... ; sweep = 0: a wave writes all pieces of its 64 (32) rows back to back
template <int MODE> __global__ __launch_bounds__(256) void wr(char *buf, int64_t nrows, int S, int sweep, int nt)
{
    const int pieces = S / 32;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    const v4f val = {1.f, 2.f, 3.f, (float)threadIdx.x};
    const int64_t rows_per_iter = MODE == 0 ? nth : nth / 2;
    if (sweep) {
        for (int p = 0; p < pieces; ++p)
            for (int64_t r0 = 0; r0 < nrows; r0 += rows_per_iter) {
                const int64_t r = r0 + (MODE == 0 ? gid : gid / 2);
                if (r >= nrows) continue;
                char *q = buf + r * S + 32 * p;
                if (MODE == 0) {
                    if (nt) { __builtin_nontemporal_store(val, (v4f *)q); __builtin_nontemporal_store(val, (v4f *)(q + 16)); }
                    else { *(v4f *)q = val; *(v4f *)(q + 16) = val; }
                } else {
                    if (nt) __builtin_nontemporal_store(val, (v4f *)(q + 16 * (gid & 1)));
                    else *(v4f *)(q + 16 * (gid & 1)) = val;
                }
            }
    } else {
        for (int64_t r0 = 0; r0 < nrows; r0 += rows_per_iter) {
            const int64_t r = r0 + (MODE == 0 ? gid : gid / 2);
            if (r >= nrows) continue;
            for (int p = 0; p < pieces; ++p) {
                char *q = buf + r * S + 32 * p;
                if (MODE == 0) {
                    if (nt) { __builtin_nontemporal_store(val, (v4f *)q); __builtin_nontemporal_store(val, (v4f *)(q + 16)); }
                    else { *(v4f *)q = val; *(v4f *)(q + 16) = val; }
                } else {
                    if (nt) __builtin_nontemporal_store(val, (v4f *)(q + 16 * (gid & 1)));
                    else *(v4f *)(q + 16 * (gid & 1)) = val;
                }
            }
        }
    }
}

int main()
{
    const int64_t bytes = (int64_t)512 << 20;
    char *buf;
    hipMalloc((void **)&buf, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 2048;
    for (int S : {32, 64, 96, 128, 256, 512})
        for (int mode = 0; mode < 2; ++mode)
            for (int sweep = 0; sweep < 2; ++sweep)
                for (int nt = 0; nt < 2; ++nt) {
                    const int64_t nrows = bytes / S;
                    for (int it = 0; it < 60; ++it) {
                        if (it == 10) hipEventRecord(e0);
                        if (mode == 0) hipLaunchKernelGGL(wr<0>, dim3(grid), dim3(256), 0, 0, buf, nrows, S, sweep, nt);
                        else hipLaunchKernelGGL(wr<1>, dim3(grid), dim3(256), 0, 0, buf, nrows, S, sweep, nt);
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    ms /= 50;
                    printf("row %3d B  %s  %s  %s: %.4f ms  %.2f TB/s\n", S, mode ? "lane pair per 32-byte piece" : "lane per 32-byte piece     ",
                           sweep ? "sweep (pieces far apart in time)" : "burst (a row's pieces together) ", nt ? "nontemporal" : "plain      ", ms, bytes / ms / 1e9);
                }
    return 0;
}
