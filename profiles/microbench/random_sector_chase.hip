// Microbenchmark: dependent random 8-byte reads from a table far larger than the caches --
// the access pattern of the LF walk (k_invert) without any other work.  Gives the achievable
// rate of random 64-byte sectors on this GPU for 1 and 2 independent chains per thread.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
__device__ __host__ inline u64 mix64(u64 x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void fill(u64 *tab, u64 n) { for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) tab[i] = mix64(i) % n; }
template <int CH, int NT>
__global__ void chase(const u64 *__restrict__ tab, u64 n, u64 walks, int steps, u64 *out)
{
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * CH; i < walks; i += (u64)gridDim.x * blockDim.x * CH) {
        u64 j[CH], acc = 0;
        for (int c = 0; c < CH; c++) j[c] = mix64(i + c) % n;
        for (int s = 0; s < steps; s++)
            for (int c = 0; c < CH; c++) { j[c] = NT ? __builtin_nontemporal_load(tab + j[c]) : tab[j[c]]; acc += j[c]; }
        out[i / CH] = acc;
    }
}
template <int CH, int NT> float run(const u64 *tab, u64 n, u64 walks, int steps, u64 *out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    chase<CH, NT><<<1 << 17, 256>>>(tab, n, walks, steps, out);   // warm
    hipEventRecord(a);
    chase<CH, NT><<<1 << 17, 256>>>(tab, n, walks, steps, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    u64 n = 4530000000ull, walks = 30000000ull; int steps = 150;
    u64 *tab, *out;
    if (hipMalloc(&tab, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, walks * 8);
    fill<<<1 << 16, 256>>>(tab, n); hipDeviceSynchronize();
    double sect = (double)walks * steps * 64 / 1e9;
    float t;
    t = run<1, 0>(tab, n, walks, steps, out); printf("1 chain  plain : %.1f ms  %.2f Gstep/s  %.2f TB/s of 64-B sectors\n", t, walks * (double)steps / t / 1e6, sect / t);
    t = run<1, 1>(tab, n, walks, steps, out); printf("1 chain  nt    : %.1f ms  %.2f Gstep/s  %.2f TB/s\n", t, walks * (double)steps / t / 1e6, sect / t);
    t = run<2, 1>(tab, n, walks, steps, out); printf("2 chains nt    : %.1f ms  %.2f Gstep/s  %.2f TB/s\n", t, walks * (double)steps / t / 1e6, sect / t);
    t = run<4, 1>(tab, n, walks, steps, out); printf("4 chains nt    : %.1f ms  %.2f Gstep/s  %.2f TB/s\n", t, walks * (double)steps / t / 1e6, sect / t);
    return 0;
}
