// k_invert.hip -- step 4: LF-mapping inversion of the (edited) eBWT back to reads.
//
// Restates invert() of bfq_int.cpp:748-819 as a batch of independent walks, one
// per read (the lock-step formulation of bfq_ext's BCRdecode, decode.cpp:499-686):
// read i starts at row i; each step emits (replacement or eBWT symbol, quality
// -- Illumina-binned when B=1, bfq_int.cpp:784-786) and moves to LF(row) until
// the terminator row.  One step = ONE 8-byte read of the LF table (bfq_rank.h);
// N walks in flight hide the dependent-load latency; output bytes are collected in
// registers and stored 16 at a time.
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

// bytes [from,to) of the 16-byte chunk at from & ~15 (lo = bytes 0..7, hi = bytes 8..15)
__device__ __forceinline__ void flush_bytes(u8 *dst, u64 from, u64 to, u64 lo, u64 hi)
{
    for (u64 p = from; p < to; p++) {
        u32 o = (u32)p & 15u;
        dst[p] = (u8)((o < 8 ? lo : hi) >> (8 * (o & 7u)));
    }
}

template <int NT>
__global__ __launch_bounds__(256) void k_invert(RankIndex R, u64 N, const u64 *__restrict__ roff, int B,
                                                u8 *__restrict__ out_bases, u8 *__restrict__ out_quals, DevCounters *cnt)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        const u64 lo = roff[i], end = roff[i + 1];
        u64 pos = end, j = i;
        u64 bl = 0, bh = 0, ql = 0, qh = 0;                            // 16 output bytes of each stream
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
            u32 o = (u32)pos & 15u, sh = (o & 7u) * 8u;
            if (o < 8) { bl |= (u64)sym << sh; ql |= (u64)q << sh; }
            else { bh |= (u64)sym << sh; qh |= (u64)q << sh; }
            if (o == 0) {                                            // chunk [pos,pos+16) is complete or clipped by `end`
                if (pos + 16 <= end) {
                    *(ulonglong2 *)(out_bases + pos) = make_ulonglong2(bl, bh);
                    *(ulonglong2 *)(out_quals + pos) = make_ulonglong2(ql, qh);
                } else {
                    flush_bytes(out_bases, pos, end, bl, bh);
                    flush_bytes(out_quals, pos, end, ql, qh);
                }
                bl = bh = ql = qh = 0;
            }
            j = nx;
        }
        if (!bad && (lo & 15)) {                                     // leading partial chunk [lo, min(end, align_up(lo)))
            u64 hi = (lo + 15) & ~15ull;
            if (hi > end) hi = end;
            flush_bytes(out_bases, lo, hi, bl, bh);
            flush_bytes(out_quals, lo, hi, ql, qh);
        }
        if (!bad && lfq_code(R.lfq[j]) != 0) bad = true;             // read longer than its slot
        if (bad) atomicAdd(&cnt->errInvert, 1ull);
    }
}

void bfq_invert_count(bfq_ctx *c, const RankIndex &R, u64 N, u32 *lens)
{
    if (!N) return;
    KLAUNCH(c, K_INVERT_COUNT, 64.0 * (double)(R.n - N), k_invert_count, bfq_grid(N, 256), 256, R, N, lens, c->d_cnt);
}

void bfq_invert(bfq_ctx *c, const RankIndex &R, u64 N, const u64 *d_roff, int B, u8 *out_bases, u8 *out_quals)
{
    if (!N) return;
    static const int nt = getenv("BFQ_INVERT_NT") ? atoi(getenv("BFQ_INVERT_NT")) : 1;   // nt loads: -15% (L1 bypass)
    if (nt)
        KLAUNCH(c, K_INVERT, 68.0 * (double)(R.n - N), k_invert<1>, bfq_grid(N, 256), 256, R, N, d_roff, B, out_bases, out_quals,
                c->d_cnt);
    else
        KLAUNCH(c, K_INVERT, 68.0 * (double)(R.n - N), k_invert<0>, bfq_grid(N, 256), 256, R, N, d_roff, B, out_bases, out_quals,
                c->d_cnt);
}
