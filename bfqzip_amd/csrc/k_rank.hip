// k_rank.hip -- builds the rank structure (bfq_rank.h) from the eBWT bytes.
// Replaces dna_string_n construction + build_rank_support
// (external/bwt2lcp/dna_string_n.hpp:52-109,247-285) and the F-array loop of
// dna_bwt_n.hpp:46-61.  Forbidden symbols raise errSymbol (dna_string_n.hpp:87-93).
#include "bfq_internal.h"
#include "bfq_device.h"

// one workgroup = one block of 256 rows; one wave = one group of 64 rows
__global__ __launch_bounds__(256) void k_rank_build(const u8 *__restrict__ bwt, u64 n, u32 term, RankBlock *__restrict__ blk,
                                                    u32 *__restrict__ bcnt, u64 nblk, DevCounters *cnt)
{
    __shared__ u32 wc[4][6];
    const u32 lane = bfq_lane(), w = threadIdx.x >> 6;
    for (u64 b = blockIdx.x; b < nblk; b += gridDim.x) {
        u64 r = b * 256 + threadIdx.x;
        u32 code = 7;                               // rows past the end match no symbol
        if (r < n) {
            u8 ch = bwt[r];
            code = (ch == (u8)term) ? 0u : bfq_base_code(ch);
            if (code == BFQ_CODE_INVALID) { atomicAdd(&cnt->errSymbol, 1ull); code = 4; }
        }
        u64 p0 = __ballot(code & 1u), p1 = __ballot(code & 2u), p2 = __ballot(code & 4u);
        if (lane == 0) {
            blk[b].pl[w][0] = p0;
            blk[b].pl[w][1] = p1;
            blk[b].pl[w][2] = p2;
        }
        if (lane < 6) {
            u64 m0 = (lane & 1u) ? p0 : ~p0, m1 = (lane & 2u) ? p1 : ~p1, m2 = (lane & 4u) ? p2 : ~p2;
            wc[w][lane] = (u32)__popcll(m0 & m1 & m2);
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            u32 t = threadIdx.x;
            bcnt[(u64)t * nblk + b] = wc[0][t] + wc[1][t] + wc[2][t] + wc[3][t];
        }
        __syncthreads();
    }
}

// scanned[c*nblk + b] = occurrences of code c before block b
__global__ __launch_bounds__(256) void k_rank_final(RankBlock *__restrict__ blk, u64 *__restrict__ cntN,
                                                    const u64 *__restrict__ scanned, u64 nblk, DevCounters *cnt,
                                                    u64 *__restrict__ F)
{
    for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (u64)gridDim.x * blockDim.x) {
        blk[b].cnt[0] = scanned[1 * nblk + b];
        blk[b].cnt[1] = scanned[2 * nblk + b];
        blk[b].cnt[2] = scanned[3 * nblk + b];
        blk[b].cnt[3] = scanned[5 * nblk + b];
        cntN[b] = scanned[4 * nblk + b];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u64 acc = 0;
        for (int s = 0; s < 6; s++) { F[s] = acc; acc += cnt->tot[s]; }
    }
}

RankIndex bfq_rank_build(bfq_ctx *c, const u8 *bwt, u64 n, int term)
{
    RankIndex R;
    u64 nblk = n / 256 + 1;
    RankBlock *blk = c->alloc<RankBlock>(nblk);
    u64 *cntN = c->alloc<u64>(nblk);
    u64 *F = c->alloc<u64>(8);
    size_t m = c->mark();
    u32 *bcnt = c->alloc<u32>(6 * nblk);
    u64 *scanned = c->alloc<u64>(6 * nblk);
    KLAUNCH(c, K_RANK_BUILD, (double)n + 0.5 * (double)n, k_rank_build, bfq_grid(nblk, 1), 256, bwt, n, (u32)(term & 0xFF), blk, bcnt,
            nblk, c->d_cnt);
    for (int s = 0; s < 6; s++) bfq_exscan_u32(c, bcnt + (u64)s * nblk, scanned + (u64)s * nblk, nblk, &c->d_cnt->tot[s]);
    KLAUNCH(c, K_RANK_FINAL, 80.0 * (double)nblk, k_rank_final, bfq_grid(nblk, 256), 256, blk, cntN,
            (const u64 *)scanned, nblk, c->d_cnt, F);
    c->release(m);
    R.blk = blk; R.cntN = cntN; R.F = F; R.n = n;
    return R;
}
