// k_synth.hip -- device side of the seeded synthetic-read generator (bfq_synth.h).
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_synth.h"

__global__ __launch_bounds__(256) void k_synth_lens(bfq_synth s, u32 *__restrict__ lens)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < s.N; i += (u64)gridDim.x * blockDim.x)
        lens[i] = bfq_synth_len(&s, s.first + i);
}

// one wave per read
__global__ __launch_bounds__(256) void k_synth_reads(bfq_synth s, const u64 *__restrict__ roff, u8 *__restrict__ bases,
                                                     u8 *__restrict__ quals)
{
    u32 lane = bfq_lane();
    u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 i = wave; i < s.N; i += nwaves) {
        u64 b = roff[i];
        u32 len = (u32)(roff[i + 1] - b);
        for (u32 k = lane; k < len; k += 64) {
            u8 bb, qq;
            bfq_synth_base(&s, s.first + i, len, k, &bb, &qq);
            bases[b + k] = bb;
            quals[b + k] = qq;
        }
    }
}

void bfq_synth_launch(bfq_ctx *c, const bfq_synth *s, u8 *d_bases, u8 *d_quals, u64 *d_roff)
{
    if (!s->N) { HIP_CHECK(hipMemsetAsync(d_roff, 0, sizeof(u64), c->stream)); return; }
    size_t m = c->mark();
    u32 *lens = c->alloc<u32>(s->N);
    KLAUNCH(c, K_SYNTH, 4.0 * (double)s->N, k_synth_lens, bfq_grid(s->N, 256), 256, *s, lens);
    bfq_exscan_u32(c, lens, d_roff, s->N, d_roff + s->N);
    u64 waves = s->N < (1u << 18) ? s->N : (1u << 18);
    KLAUNCH(c, K_SYNTH, 2.0 * (double)s->N * s->Lmax, k_synth_reads, ceil_div(waves, 4), 256, *s, (const u64 *)d_roff,
            d_bases, d_quals);
    c->release(m);
}

// header lines "@SYN.<number of the read in its collection, from 1>\n" as one text (what bfq_int -H reads)
__device__ __forceinline__ u32 dec_digits(u64 v) { u32 d = 1; while (v >= 10) { v /= 10; d++; } return d; }
__global__ __launch_bounds__(256) void k_synth_hdr_sizes(u64 first, u64 N, u32 *__restrict__ sizes)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x)
        sizes[i] = 6 + dec_digits(first + i + 1);
}
__global__ __launch_bounds__(256) void k_synth_hdr_write(u64 first, u64 N, const u64 *__restrict__ off, u8 *__restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        u8 *p = out + off[i];
        p[0] = '@'; p[1] = 'S'; p[2] = 'Y'; p[3] = 'N'; p[4] = '.';
        u64 v = first + i + 1;
        const u32 d = dec_digits(v);
        for (u32 k = 0; k < d; k++) { p[4 + d - k] = (u8)('0' + v % 10); v /= 10; }
        p[5 + d] = 10;
    }
}
u8 *bfq_synth_headers(bfq_ctx *c, const bfq_synth *s, u64 *len)
{
    u32 *sizes = c->alloc<u32>(s->N + 1);
    u64 *off = c->alloc<u64>(s->N + 2);
    if (s->N) KLAUNCH(c, K_SYNTH, 4.0 * (double)s->N, k_synth_hdr_sizes, bfq_grid(s->N, 256), 256, s->first, s->N, sizes);
    bfq_exscan_u32(c, sizes, off, s->N, off + s->N);
    u64 hl = 0;
    HIP_CHECK(hipMemcpyAsync(&hl, off + s->N, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    u8 *hdr = c->alloc<u8>(hl + 64);
    if (s->N) KLAUNCH(c, K_SYNTH, (double)hl, k_synth_hdr_write, bfq_grid(s->N, 256), 256, s->first, s->N, (const u64 *)off, hdr);
    *len = hl;
    return hdr;
}
