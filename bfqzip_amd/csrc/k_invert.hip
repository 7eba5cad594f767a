// k_invert.hip -- step 4: LF-mapping inversion of the (edited) eBWT back to reads.
//
// Restates invert() of bfq_int.cpp:748-819 as a batch of independent walks, one
// per read (the lock-step formulation of bfq_ext's BCRdecode, decode.cpp:499-686):
// read i starts at row i; each step emits (replacement or eBWT symbol, quality
// -- Illumina-binned when B=1, bfq_int.cpp:784-786) and moves to LF(row) until
// the terminator row.  One step = ONE 8-byte read of the LF table (bfq_rank.h);
// N walks in flight hide the dependent-load latency; output bytes are collected in
// registers and stored 16 at a time (chunks counted from the end of the read).
#include <stdlib.h>
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_rank.h"

__global__ __launch_bounds__(256) void k_invert_count(RankIndex R, u64 N, u32 *__restrict__ lens, DevCounters *cnt)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        u64 j = i;
        u32 len = 0;
        for (;;) {
            u64 x = R.lfq[j];
            if (!lfq_code(x)) break;
            u64 nx = lfq_next(x);
            if (++len > BFQ_MAX_READ_LEN || nx >= R.n) { atomicAdd(&cnt->errInvert, 1ull); break; }
            j = nx;
        }
        lens[i] = len;
    }
}

// the low `cnt` (< 8) bytes of v to dst, as 4 + 2 + 1 byte stores (unaligned stores are exact on gfx950)
__device__ __forceinline__ void store_tail(u8 *dst, u64 v, u32 cnt)
{
    if (cnt & 4u) { *(u32 *)dst = (u32)v; v >>= 32; dst += 4; }
    if (cnt & 2u) { *(u16 *)dst = (u16)v; v >>= 16; dst += 2; }
    if (cnt & 1u) *dst = (u8)v;
}

// The walk produces a read back to front.  Output bytes are collected in chunks of 16 counted from the
// read's END ([end-16, end), [end-32, end-16), ...), each stored as two unaligned 8-byte words, so only the
// front of the read (len mod 16 bytes) needs narrower stores.
// NL: the outputs are the line streams of BFQzip.py --m2/--m3 (OUT.fq.dna / OUT.fq.qs: read i at roff[i] + i, followed
// by a newline) instead of the reads back to back.  Reads [first, first + count).
template <int NT, int NL>
__global__ __launch_bounds__(256) void k_invert(RankIndex R, u64 first, u64 count, const u64 *__restrict__ roff, int B,
                                                u8 *__restrict__ out_bases, u8 *__restrict__ out_quals, DevCounters *cnt)
{
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (u64)gridDim.x * blockDim.x) {
        const u64 i = first + t;
        const u64 lo = roff[i] + (NL ? i : 0), end = roff[i + 1] + (NL ? i : 0);
        if (NL) { out_bases[end] = 10; out_quals[end] = 10; }
        u64 pos = end, j = i;
        u64 bl = 0, bh = 0, ql = 0, qh = 0;                            // 16 output bytes of each stream
        u32 have = 0;                                                  // bytes in the current chunk (filled from the top)
        bool bad = false;
        while (pos > lo) {
            u64 x = NT ? __builtin_nontemporal_load(R.lfq + j) : R.lfq[j];
            u32 code = lfq_code(x);
            u64 nx = lfq_next(x);
            if (!code || nx >= R.n) { bad = true; break; }           // walk ended before the read did
            u32 sym = bfq_code_sym(lfq_replaced(x) ? lfq_repl(x) : code);
            u32 q = lfq_qual(x);
            if (B) q = bfq_bin8(q);
            --pos;
            u32 o = 15u - have, sh = (o & 7u) * 8u;
            if (o < 8) { bl |= (u64)sym << sh; ql |= (u64)q << sh; }
            else { bh |= (u64)sym << sh; qh |= (u64)q << sh; }
            if (++have == 16) {                                      // chunk [pos, pos+16) complete
                *(u64 *)(out_bases + pos) = bl; *(u64 *)(out_bases + pos + 8) = bh;
                *(u64 *)(out_quals + pos) = ql; *(u64 *)(out_quals + pos + 8) = qh;
                bl = bh = ql = qh = 0;
                have = 0;
            }
            j = nx;
        }
        if (!bad && have) {                                          // front of the read: `have` bytes at the top of the chunk
            const u32 s = 16u - have;                                // bytes to shift out (1..15)
            u64 b0, b1, q0, q1;
            if (s >= 8) { b0 = bh >> (8 * (s - 8)); q0 = qh >> (8 * (s - 8)); b1 = q1 = 0; }
            else { b0 = (bl >> (8 * s)) | (bh << (64 - 8 * s)); q0 = (ql >> (8 * s)) | (qh << (64 - 8 * s)); b1 = bh >> (8 * s); q1 = qh >> (8 * s); }
            if (have >= 8) {
                *(u64 *)(out_bases + lo) = b0; *(u64 *)(out_quals + lo) = q0;
                store_tail(out_bases + lo + 8, b1, have - 8); store_tail(out_quals + lo + 8, q1, have - 8);
            } else {
                store_tail(out_bases + lo, b0, have); store_tail(out_quals + lo, q0, have);
            }
        }
        if (!bad && lfq_code(R.lfq[j]) != 0) bad = true;             // read longer than its slot
        if (bad) atomicAdd(&cnt->errInvert, 1ull);
    }
}

__global__ __launch_bounds__(256) void k_fixed_offsets(u64 N, u64 L, u64 *__restrict__ roff)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i <= N; i += (u64)gridDim.x * blockDim.x) roff[i] = i * L;
}
void bfq_fixed_offsets(bfq_ctx *c, u64 N, u64 L, u64 *d_roff)
{
    KLAUNCH(c, K_MISC, 8.0 * (double)N, k_fixed_offsets, bfq_grid(N + 1, 256), 256, N, L, d_roff);
}

void bfq_invert_count(bfq_ctx *c, const RankIndex &R, u64 N, u32 *lens)
{
    if (!N) return;
    KLAUNCH(c, K_INVERT_COUNT, 64.0 * (double)(R.n - N), k_invert_count, bfq_grid(N, 256), 256, R, N, lens, c->d_cnt);
}

void bfq_invert(bfq_ctx *c, const RankIndex &R, u64 N, const u64 *d_roff, int B, u8 *out_bases, u8 *out_quals, u64 first, u64 count,
                bool lines)
{
    if (count == ~0ull) count = N - first;
    if (!count) return;
    const int nt = c->env.invertNt;                      // nontemporal loads: -15% (L1 bypass)
    const double bytes = 68.0 * (double)(R.n - N) * ((double)count / (double)N);
#define INV_LAUNCH(NTV, NLV) KLAUNCH(c, K_INVERT, bytes, (k_invert<NTV, NLV>), bfq_grid(count, 256), 256, R, first, count, d_roff, B, out_bases, out_quals, c->d_cnt)
    if (nt) { if (lines) INV_LAUNCH(1, 1); else INV_LAUNCH(1, 0); }
    else { if (lines) INV_LAUNCH(0, 1); else INV_LAUNCH(0, 0); }
#undef INV_LAUNCH
}
