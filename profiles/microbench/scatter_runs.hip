// scatter_runs.hip -- what the memory system gives a radix-scatter's access pattern, with no ranking at all:
// records (u32 + u64, as in k_radix_scatter) are read in order and written in runs of R consecutive records to
// pseudo-randomly permuted places.  A stable 8-bit LSD pass over 3072-record tiles writes runs of ~12 records.
//   hipcc -O3 --offload-arch=gfx950 scatter_runs.hip -o scatter_runs && ./scatter_runs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// run r of length R goes to slot perm(r); perm = multiplication by an odd constant modulo 2^k (a bijection)
__global__ __launch_bounds__(256) void k_scatter(const u32 *__restrict__ a0, const u64 *__restrict__ a12, u32 *__restrict__ b0,
                                                 u64 *__restrict__ b12, u64 n, u32 R, u64 runMask)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 run = i / R, off = i - run * R;
        u64 dst = ((run * 0x9E3779B97F4A7C15ull) & runMask) * R + off;
        b0[dst] = a0[i];
        b12[dst] = a12[i];
    }
}

int main()
{
    const u64 n = 1ull << 30;                                    // 1 G records, 12 GB in + 12 GB out
    u32 *a0, *b0; u64 *a12, *b12;
    CK(hipMalloc(&a0, 4 * n)); CK(hipMalloc(&b0, 4 * n * 2)); CK(hipMalloc(&a12, 8 * n)); CK(hipMalloc(&b12, 8 * n * 2));
    CK(hipMemset(a0, 1, 4 * n)); CK(hipMemset(a12, 2, 8 * n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u32 Rs[] = {4, 8, 12, 16, 24, 48, 96, 1024};
    for (u32 R : Rs) {
        u64 runs = 1; while (runs * R < n) runs <<= 1;           // power of two >= n / R (output buffers are 2x)
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0)); k_scatter<<<1u << 16, 256>>>(a0, a12, b0, b12, n, R, runs - 1); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("runs of %4u records (%5u B of w0, %5u B of w12): %7.2f ms  %6.0f GB/s (24 B per record)\n", R, 4 * R, 8 * R, ms,
               24.0 * n / ms / 1e6);
    }
    return 0;
}
