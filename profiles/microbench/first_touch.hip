// first_touch.hip -- how long does the first allocation of HBM that has not been used since boot take, and does it help to ask
// for it in pieces from several threads?  (bfq_int's 84 GiB arena costs 1.93 s on a fresh box.)
//   ./first_touch <GiB per thread> <threads> [vmm | free <seconds>]
// vmm: hipMemCreate chunks (one per thread) mapped into one address range instead of separate hipMalloc calls.
// free <s>: hipFree everything, then sleep <s> seconds before the process ends (is freed memory scrubbed at hipFree or at exit?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <thread>
#include <vector>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? atoll(argv[1]) : 8;
    const int T = argc > 2 ? atoi(argv[2]) : 1;
    const bool vmm = argc > 3 && !strcmp(argv[3], "vmm");
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    std::vector<void *> p(T, nullptr);
    std::vector<double> dt(T, 0);
    const size_t bytes = gib << 30;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    std::vector<hipMemGenericAllocationHandle_t> h(T);
    void *va = nullptr;
    if (vmm && hipMemAddressReserve(&va, bytes * T, 1ull << 30, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
    const double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            (void)hipSetDevice(0);
            const double a = now();
            hipError_t e = vmm ? hipMemCreate(&h[t], bytes, &prop, 0) : hipMalloc(&p[t], bytes);
            if (e == hipSuccess && vmm) e = hipMemMap((char *)va + bytes * t, bytes, 0, h[t], 0);
            dt[t] = now() - a;
            if (e != hipSuccess) printf("thread %d: %s\n", t, hipGetErrorString(e));
        });
    for (auto &x : th) x.join();
    const double all = now() - t0;
    printf("%s: %d thread(s) x %zu GiB: %.3f s in all = %.1f GB/s (per thread:", vmm ? "hipMemCreate+Map" : "hipMalloc", T, gib, all, bytes * T / 1e9 / all);
    for (int t = 0; t < T; t++) printf(" %.2f", dt[t]);
    printf(")\n");
    if (argc > 4 && !strcmp(argv[3], "free")) {
        const double f0 = now();
        for (int t = 0; t < T; t++) (void)hipFree(p[t]);
        (void)hipDeviceSynchronize();
        printf("hipFree: %.3f s; sleeping %s s\n", now() - f0, argv[4]);
        struct timespec ts = {atoi(argv[4]), 0};
        nanosleep(&ts, nullptr);
    }
    return 0;
}
