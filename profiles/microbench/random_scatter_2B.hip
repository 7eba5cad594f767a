// Microbenchmark for VERDICT r1 item 6a: in the fused path the sort payload carries the text position of every
// row, so the inversion could SCATTER (symbol, quality) to out[position] instead of walking the LF table.
// Pattern: rows read in order (8-byte position + 2 payload bytes), one 2-byte store per row to a pseudo-random
// place of a 9 GB array (a permutation: every 2-byte slot written exactly once), against the LF walk's
// 84 ms / 4.5 G dependent random 8-byte reads (random_sector_chase.hip).
// Variants: 2-byte stores, nontemporal 2-byte stores, 1-byte stores into two arrays (bases / quals separately).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
typedef unsigned short u16;
typedef unsigned char u8;
// bijection on [0, 2^k): multiply by an odd constant, xor-shift, multiply (all invertible mod 2^k)
__device__ inline u64 perm(u64 x, int k) { u64 m = (1ull << k) - 1; x = (x * 0x9E3779B97F4A7C15ull) & m; x ^= x >> (k / 2); x = (x * 0xBF58476D1CE4E5B9ull) & m; x ^= x >> (k / 2 + 1); return x & m; }
__global__ void mkpos(u64 *pos, u64 n, int k) { for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) pos[i] = perm(i, k); }
template <int MODE>
__global__ void scatter(const u64 *__restrict__ pos, const u16 *__restrict__ val, u64 n, u16 *__restrict__ out, u8 *__restrict__ o1, u8 *__restrict__ o2)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 p = pos[i];
        const u16 v = val[i];
        if (MODE == 0) out[p] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, out + p);
        else { o1[p] = (u8)v; o2[p] = (u8)(v >> 8); }
    }
}
template <int MODE> float run(const u64 *pos, const u16 *val, u64 n, u16 *out, u8 *o1, u8 *o2)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    scatter<MODE><<<1 << 18, 256>>>(pos, val, n, out, o1, o2);
    hipEventRecord(a);
    scatter<MODE><<<1 << 18, 256>>>(pos, val, n, out, o1, o2);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    const int k = 32; const u64 n = 1ull << k;            // 4.29 G rows (30 M x 150 has 4.53 G)
    u64 *pos; u16 *val, *out; u8 *o1, *o2;
    if (hipMalloc(&pos, n * 8) != hipSuccess || hipMalloc(&val, n * 2) != hipSuccess || hipMalloc(&out, n * 2) != hipSuccess ||
        hipMalloc(&o1, n) != hipSuccess || hipMalloc(&o2, n) != hipSuccess) { printf("alloc failed\n"); return 1; }
    mkpos<<<1 << 16, 256>>>(pos, n, k); hipMemset(val, 1, n * 2); hipDeviceSynchronize();
    float t;
    t = run<0>(pos, val, n, out, o1, o2); printf("2-byte stores        : %.1f ms  %.2f Gstore/s  (x 4.53/4.29 = %.1f ms at 30Mx150)\n", t, n / t / 1e6, t * 4.53 / 4.295);
    t = run<1>(pos, val, n, out, o1, o2); printf("2-byte nt stores     : %.1f ms  %.2f Gstore/s  (%.1f ms)\n", t, n / t / 1e6, t * 4.53 / 4.295);
    t = run<2>(pos, val, n, out, o1, o2); printf("2 x 1-byte stores    : %.1f ms  %.2f Grow/s    (%.1f ms)\n", t, n / t / 1e6, t * 4.53 / 4.295);
    return 0;
}
