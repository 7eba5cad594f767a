// stream_ceilings.hip -- what plain streaming kernels reach on this chip: the yardstick for the
// streaming stages (k_build_keys, k_emit_bwt, k_lf_build, the radix passes).
//   hipcc -O3 --offload-arch=gfx950 stream_ceilings.hip -o stream_ceilings && ./stream_ceilings
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ in, u64 n16, u64 *sink)
{
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) {
        uint4 v = in[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x123456789ull) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4 *__restrict__ out, u64 n16)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x)
        out[i] = make_uint4((unsigned)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n16)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) out[i] = in[i];
}
// 2 bytes in, 8 bytes out per element: the shape of k_lf_build
__global__ __launch_bounds__(256) void k_expand(const unsigned char *__restrict__ a, const unsigned char *__restrict__ b,
                                                u64 *__restrict__ out, u64 n)
{
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g * 4 < n; g += (u64)gridDim.x * blockDim.x) {
        unsigned x = *(const unsigned *)(a + 4 * g), y = *(const unsigned *)(b + 4 * g);
        u64 o[4];
        for (int k = 0; k < 4; k++) o[k] = ((u64)((x >> (8 * k)) & 255) << 40) | ((u64)((y >> (8 * k)) & 255) << 48) | (g * 4 + k);
        ((uint4 *)out)[2 * g] = make_uint4((unsigned)o[0], (unsigned)(o[0] >> 32), (unsigned)o[1], (unsigned)(o[1] >> 32));
        ((uint4 *)out)[2 * g + 1] = make_uint4((unsigned)o[2], (unsigned)(o[2] >> 32), (unsigned)o[3], (unsigned)(o[3] >> 32));
    }
}

int main()
{
    const u64 bytes = 32ull << 30, n16 = bytes / 16;
    uint4 *a, *b; u64 *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grids[3] = {256 * 8, 256 * 32, 1u << 19};
    for (int gi = 0; gi < 3; gi++) {
        unsigned g = grids[gi];
        float ms;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0)); k_read<<<g, 256>>>(a, n16, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("grid %7u  read  %6.2f ms  %6.0f GB/s\n", g, ms, bytes / ms / 1e6);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0)); k_write<<<g, 256>>>(b, n16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("grid %7u  write %6.2f ms  %6.0f GB/s\n", g, ms, bytes / ms / 1e6);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0)); k_copy<<<g, 256>>>(a, b, n16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("grid %7u  copy  %6.2f ms  %6.0f GB/s (read+write)\n", g, ms, 2.0 * bytes / ms / 1e6);
        u64 n = bytes / 8;     // elements: 2 B in, 8 B out
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0)); k_expand<<<g, 256>>>((const unsigned char *)a, (const unsigned char *)a + n, (u64 *)b, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("grid %7u  2B->8B %6.2f ms  %6.0f GB/s (10 B/elem)\n", g, ms, 10.0 * n / ms / 1e6);
    }
    return 0;
}
