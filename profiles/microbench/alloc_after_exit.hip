// Microbenchmark behind DESIGN 4c's remark on hipMalloc: how long does a process wait for HBM right after another process
// that held ~100 GiB has exited?   usage: alloc_after_exit hold <GiB>   (allocate, touch, exit)
//                                         alloc_after_exit big <GiB>    (time ONE hipMalloc)
//                                         alloc_after_exit chunks <GiB> (time <GiB> hipMallocs of 1 GiB)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    size_t gib = (size_t)atol(argv[2]);
    double t0 = now();
    hipFree(0);
    double t1 = now();
    if (!strcmp(argv[1], "hold")) {
        void *p; if (hipMalloc(&p, gib << 30) != hipSuccess) { printf("hold: alloc failed\n"); return 1; }
        hipMemset(p, 1, gib << 30); hipDeviceSynchronize();
        printf("hold: init %.3f s, %zu GiB allocated and touched in %.3f s\n", t1 - t0, gib, now() - t1);
        return 0;                                   // exit without freeing: the driver cleans up
    }
    if (!strcmp(argv[1], "big")) {
        void *p; hipError_t e = hipMalloc(&p, gib << 30);
        printf("big: init %.3f s, hipMalloc(%zu GiB) %s in %.3f s\n", t1 - t0, gib, e == hipSuccess ? "ok" : "FAILED", now() - t1);
        return 0;
    }
    double worst = 0;
    for (size_t i = 0; i < gib; i++) { void *p; double a = now(); if (hipMalloc(&p, 1ull << 30) != hipSuccess) { printf("chunk %zu failed\n", i); break; } double d = now() - a; if (d > worst) worst = d; }
    printf("chunks: init %.3f s, %zu x hipMalloc(1 GiB) in %.3f s (slowest %.3f s)\n", t1 - t0, gib, now() - t1, worst);
    return 0;
}
