// k_scan.hip -- device-wide exclusive prefix sums (reduce, scan the partials, scan down).
// Used for radix-sort offsets, rank-structure counters and compactions.
#include "bfq_internal.h"
#include "bfq_device.h"

#define SC_THREADS 256
#define SC_ITEMS 16
#define SC_CHUNK (SC_THREADS * SC_ITEMS)

template <class T>
__global__ __launch_bounds__(SC_THREADS) void k_scan_reduce(const T *__restrict__ in, u64 n, u64 *__restrict__ sums)
{
    __shared__ u64 sh[4];
    u64 base = (u64)blockIdx.x * SC_CHUNK;
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        u64 i = base + (u64)k * SC_THREADS + threadIdx.x;
        if (i < n) s += (u64)in[i];
    }
    u64 inc = bfq_wave_incscan64(s);
    if (bfq_lane() == 63) sh[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// each thread owns SC_ITEMS consecutive elements
template <class T>
__global__ __launch_bounds__(SC_THREADS) void k_scan_down(const T *__restrict__ in, u64 *__restrict__ out, u64 n,
                                                          const u64 *__restrict__ blockBase, u64 *__restrict__ total)
{
    __shared__ u64 sh[4];
    u64 base = (u64)blockIdx.x * SC_CHUNK + (u64)threadIdx.x * SC_ITEMS;
    u64 v[SC_ITEMS];
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        u64 i = base + k;
        v[k] = (i < n) ? (u64)in[i] : 0;
        s += v[k];
    }
    u64 tot;
    u64 ex = bfq_block_exscan64(s, sh, &tot);
    u64 run = ex + (blockBase ? blockBase[blockIdx.x] : 0);
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        u64 i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SC_THREADS - 1) *total = run;
}

template <class T>
static void exscan_impl(bfq_ctx *c, const T *in, u64 *out, u64 n, u64 *d_total)
{
    if (n == 0) {
        if (d_total) HIP_CHECK(hipMemsetAsync(d_total, 0, sizeof(u64), c->stream));
        return;
    }
    u64 nb = ceil_div(n, SC_CHUNK);
    if (nb == 1) {
        KLAUNCH(c, K_SCAN, n * (sizeof(T) + 8), (k_scan_down<T>), 1, SC_THREADS, in, out, n, (const u64 *)nullptr, d_total);
        return;
    }
    size_t m = c->mark();
    u64 *sums = c->alloc<u64>(nb);
    u64 *sumsScanned = c->alloc<u64>(nb);
    KLAUNCH(c, K_SCAN, n * sizeof(T), (k_scan_reduce<T>), nb, SC_THREADS, in, n, sums);
    exscan_impl<u64>(c, sums, sumsScanned, nb, nullptr);
    KLAUNCH(c, K_SCAN, n * (sizeof(T) + 8), (k_scan_down<T>), nb, SC_THREADS, in, out, n, (const u64 *)sumsScanned, d_total);
    c->release(m);   // stream-ordered: later kernels that reuse this space run after the scan
}

void bfq_exscan_u8(bfq_ctx *c, const u8 *in, u64 *out, u64 n, u64 *t) { exscan_impl<u8>(c, in, out, n, t); }
void bfq_exscan_u32(bfq_ctx *c, const u32 *in, u64 *out, u64 n, u64 *t) { exscan_impl<u32>(c, in, out, n, t); }
void bfq_exscan_u64(bfq_ctx *c, const u64 *in, u64 *out, u64 n, u64 *t) { exscan_impl<u64>(c, in, out, n, t); }
