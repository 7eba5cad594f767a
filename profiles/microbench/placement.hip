// placement.hip -- does WHERE two buffers lie inside one large allocation change the rate of moving data between them?
// bench.py's per-pass times of k_radix_scatter alternate (B->A 32-36 ms, A->B 26-28 ms) and flip when the two record
// buffers swap places in the arena (BFQ_AB_SWAP=1): this isolates the effect with plain kernels.
//   hipcc -O3 --offload-arch=gfx950 placement.hip -o placement && ./placement [GiB of the allocation, default 120]
// 1. read and write rate of every 8 GiB slice of the allocation;
// 2. copy 27 GiB low -> high and high -> low (streaming, 16 B per lane);
// 3. the same with the writes scattered into 256 streams (a radix pass without the ranking).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ in, u64 n16, u64 *sink)
{
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) { uint4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x123456789ull) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4 *__restrict__ out, u64 n16)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) out[i] = make_uint4((unsigned)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n16)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) out[i] = in[i];
}
// tile t of 4096 uint4 (64 KiB) is cut into 256 runs of 16 uint4 (256 B); run d of tile t goes to stream d at position t:
// out[(d * tiles + t) * 16 .. + 16) -- every workgroup writes 256 different places, like a radix pass
__global__ __launch_bounds__(256) void k_scatter256(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 tiles)
{
    for (u64 t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint4 *src = in + t * 4096;
#pragma unroll 4
        for (int k = 0; k < 16; k++) {
            const unsigned e = k * 256 + threadIdx.x;            // element of the tile
            const unsigned d = e >> 4, j = e & 15;
            out[((u64)d * tiles + t) * 16 + j] = src[e];
        }
    }
}

int main(int argc, char **argv)
{
    const u64 GiB = 1ull << 30;
    const u64 total = (argc > 1 ? strtoull(argv[1], 0, 10) : 120) * GiB;
    char *base; u64 *sink;
    CK(hipMalloc(&base, total)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(base, 1, total));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    const unsigned g = 256 * 32;
    printf("allocation %llu GiB at %p\n", total / GiB, (void *)base);
    const u64 sl = 8 * GiB;
    for (u64 o = 0; o + sl <= total; o += sl) {
        float r = 0, w = 0;
        for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0)); k_read<<<g, 256>>>((const uint4 *)(base + o), sl / 16, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&r, e0, e1)); }
        for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0)); k_write<<<g, 256>>>((uint4 *)(base + o), sl / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&w, e0, e1)); }
        printf("slice at %3llu GiB: read %5.0f GB/s  write %5.0f GB/s\n", o / GiB, sl / r / 1e6, sl / w / 1e6);
    }
    // two 27 GiB regions the way the sort's buffers lie: low at 19 GiB, high at 19 + 54 = 73 GiB (120 GiB allocation)
    const u64 reg = 27 * GiB;
    const u64 lo = total >= 110 * GiB ? 19 * GiB : 0, hi = total >= 110 * GiB ? 73 * GiB : total - reg;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); k_copy<<<g, 256>>>((const uint4 *)(base + lo), (uint4 *)(base + hi), reg / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy    low -> high: %6.2f ms  %5.0f GB/s (read + write)\n", ms, 2.0 * reg / ms / 1e6);
        CK(hipEventRecord(e0)); k_copy<<<g, 256>>>((const uint4 *)(base + hi), (uint4 *)(base + lo), reg / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy    high -> low: %6.2f ms  %5.0f GB/s\n", ms, 2.0 * reg / ms / 1e6);
    }
    const u64 tiles = reg / 65536;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); k_scatter256<<<g, 256>>>((const uint4 *)(base + lo), (uint4 *)(base + hi), tiles); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("scatter low -> high: %6.2f ms  %5.0f GB/s\n", ms, 2.0 * reg / ms / 1e6);
        CK(hipEventRecord(e0)); k_scatter256<<<g, 256>>>((const uint4 *)(base + hi), (uint4 *)(base + lo), tiles); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("scatter high -> low: %6.2f ms  %5.0f GB/s\n", ms, 2.0 * reg / ms / 1e6);
    }
    // neighbours: the two regions back to back in the middle of the allocation
    const u64 mid = (total / 2 / GiB) * GiB;
    if (mid >= reg && mid + reg <= total)
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0)); k_scatter256<<<g, 256>>>((const uint4 *)(base + mid - reg), (uint4 *)(base + mid), tiles); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("scatter (mid-27) -> mid: %6.2f ms  %5.0f GB/s\n", ms, 2.0 * reg / ms / 1e6);
            CK(hipEventRecord(e0)); k_scatter256<<<g, 256>>>((const uint4 *)(base + mid), (uint4 *)(base + mid - reg), tiles); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("scatter mid -> (mid-27): %6.2f ms  %5.0f GB/s\n", ms, 2.0 * reg / ms / 1e6);
        }
    return 0;
}
