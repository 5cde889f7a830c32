// ubench_hostcopy.hip -- what the PCIe side of the host-array API can reach on this box (developer measurement;
// results quoted in DESIGN.md section 6).  hipcc --offload-arch=gfx950 -O2 tools/ubench_hostcopy.hip -o /tmp/hc -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void *fresh(size_t bytes) {   // like np.empty for a large array: anonymous mapping, pages untouched
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}
static void prefault(char *p, size_t bytes, int nthreads) {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([=]() {
            size_t a = bytes * t / nthreads, b = bytes * (t + 1) / nthreads;
            a &= ~(size_t)4095; b = (t + 1 == nthreads) ? bytes : (b & ~(size_t)4095);
#ifdef MADV_POPULATE_WRITE
            if (madvise(p + a, b - a, MADV_POPULATE_WRITE) == 0) return;
#endif
            for (size_t i = a; i < b; i += 4096) p[i] = 0;
        });
    for (auto &t : th) t.join();
}
int main() {
    const size_t B = (size_t)512 << 20;
    void *d_in, *d_out;
    CK(hipMalloc(&d_in, B)); CK(hipMalloc(&d_out, B));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    char *x = (char *)fresh(B); memset(x, 1, B);
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now(); CK(hipMemcpy(d_in, x, B, hipMemcpyHostToDevice)); double t1 = now();
        printf("H2D pageable 512 MiB (touched source): %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, B / (t1 - t0) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        char *y = (char *)fresh(B);
        double t0 = now(); CK(hipMemcpy(y, d_out, B, hipMemcpyDeviceToHost)); double t1 = now();
        printf("D2H pageable 512 MiB into FRESH pages: %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, B / (t1 - t0) / 1e9);
        t0 = now(); CK(hipMemcpy(y, d_out, B, hipMemcpyDeviceToHost)); t1 = now();
        printf("D2H pageable 512 MiB into touched pages: %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, B / (t1 - t0) / 1e9);
        munmap(y, B);
    }
    for (int nt : {1, 2, 4, 8, 16}) {
        char *y = (char *)fresh(B);
        double t0 = now(); prefault(y, B, nt); double t1 = now();
        printf("prefault 512 MiB with %2d threads: %.1f ms  %.1f GB/s\n", nt, (t1 - t0) * 1e3, B / (t1 - t0) / 1e9);
        munmap(y, B);
    }
    {
        char *y = (char *)fresh(B);
        double t0 = now(); CK(hipHostRegister(y, B, hipHostRegisterDefault)); double t1 = now();
        printf("hipHostRegister 512 MiB FRESH: %.1f ms\n", (t1 - t0) * 1e3);
        t0 = now(); CK(hipMemcpyAsync(y, d_out, B, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); t1 = now();
        printf("D2H registered 512 MiB: %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, B / (t1 - t0) / 1e9);
        t0 = now(); CK(hipHostUnregister(y)); t1 = now();
        printf("hipHostUnregister: %.1f ms\n", (t1 - t0) * 1e3);
        t0 = now(); CK(hipHostRegister(x, B, hipHostRegisterDefault)); t1 = now();
        printf("hipHostRegister 512 MiB touched: %.1f ms\n", (t1 - t0) * 1e3);
        CK(hipHostRegister(y, B, hipHostRegisterDefault));
        t0 = now();
        CK(hipMemcpyAsync(d_in, x, B, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(y, d_out, B, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); t1 = now();
        printf("registered H2D + D2H concurrently, 512 MiB each: %.1f ms  (%.1f GB/s per direction)\n", (t1 - t0) * 1e3, B / (t1 - t0) / 1e9);
        // chunked, 32 MiB pieces alternating
        t0 = now();
        for (size_t o = 0; o < B; o += (size_t)32 << 20) {
            CK(hipMemcpyAsync((char *)d_in + o, x + o, (size_t)32 << 20, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync(y + o, (char *)d_out + o, (size_t)32 << 20, hipMemcpyDeviceToHost, s2));
        }
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); t1 = now();
        printf("same in 32 MiB pieces: %.1f ms\n", (t1 - t0) * 1e3);
        CK(hipHostUnregister(x)); CK(hipHostUnregister(y));
        munmap(y, B);
    }
    {   // two threads, pageable, opposite directions (what the first pipeline did)
        char *y = (char *)fresh(B); prefault(y, B, 8);
        double t0 = now();
        std::thread th([&]() { (void)hipSetDevice(0); (void)hipMemcpyAsync(y, d_out, B, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2); });
        CK(hipMemcpyAsync(d_in, x, B, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
        th.join();
        double t1 = now();
        printf("pageable H2D and D2H from two threads, 512 MiB each: %.1f ms\n", (t1 - t0) * 1e3);
        munmap(y, B);
    }
    return 0;
}
