// k_invert.hip -- step 4: LF-mapping inversion of the (edited) eBWT back to reads.
//
// Restates invert() of bfq_int.cpp:748-819 as a batch of independent walks, one
// per read (the lock-step formulation of bfq_ext's BCRdecode, decode.cpp:499-686):
// read i starts at row i; each step emits (replacement or eBWT symbol, quality
// -- Illumina-binned when B=1, bfq_int.cpp:784-786) and moves to LF(row) until
// the terminator row.  One step = one 128-byte rank block + one quality byte +
// one replacement byte; N walks in flight hide the dependent-load latency.
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_rank.h"

__global__ __launch_bounds__(256) void k_invert_count(RankIndex R, u64 N, u32 *__restrict__ lens, DevCounters *cnt)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        u64 j = i, nx = 0;
        u32 len = 0;
        for (;;) {
            u32 code = rank_step(R, j, &nx);
            if (!code) break;
            if (++len > BFQ_MAX_READ_LEN || nx >= R.n) { atomicAdd(&cnt->errInvert, 1ull); break; }
            j = nx;
        }
        lens[i] = len;
    }
}

__global__ __launch_bounds__(256) void k_invert(RankIndex R, const u8 *__restrict__ qual, const u8 *__restrict__ modsym,
                                                u64 N, const u64 *__restrict__ roff, int B, u8 *__restrict__ out_bases,
                                                u8 *__restrict__ out_quals, DevCounters *cnt)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
    u64 lo = roff[i], pos = roff[i + 1];
    u64 j = i, nx = 0;
    bool bad = false;
    while (pos > lo) {
        u32 code = rank_step(R, j, &nx);
        if (!code || nx >= R.n) { bad = true; break; }           // walk ended before the read did
        u8 ms = modsym ? modsym[j] : (u8)0;
        u32 q = qual[j];
        --pos;
        out_bases[pos] = ms ? ms : bfq_code_sym(code);
        out_quals[pos] = (u8)(B ? bfq_bin8(q) : q);
        j = nx;
    }
    if (!bad && rank_code_at(R, j) != 0) bad = true;             // read longer than its slot
    if (bad) atomicAdd(&cnt->errInvert, 1ull);
    }
}

void bfq_invert_count(bfq_ctx *c, const RankIndex &R, u64 N, u32 *lens)
{
    if (!N) return;
    KLAUNCH(c, K_INVERT_COUNT, 128.0 * (double)(R.n - N), k_invert_count, bfq_grid(N, 256), 256, R, N, lens, c->d_cnt);
}

void bfq_invert(bfq_ctx *c, const RankIndex &R, const u8 *qual, const u8 *modsym, u64 N, const u64 *d_roff,
                u8 *out_bases, u8 *out_quals)
{
    if (!N) return;
    KLAUNCH(c, K_INVERT, 68.0 * (double)(R.n - N), k_invert, bfq_grid(N, 256), 256, R, qual, modsym, N, d_roff,
            c->P.B, out_bases, out_quals, c->d_cnt);
}
