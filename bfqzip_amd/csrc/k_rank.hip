// k_rank.hip -- builds the rank structure (bfq_rank.h) from the eBWT bytes and the
// permuted qualities.  Replaces dna_string_n construction + build_rank_support
// (external/bwt2lcp/dna_string_n.hpp:52-109,247-285), the F-array loop of
// dna_bwt_n.hpp:46-61 and the QUAL array load of bfq_int.cpp:640-651.
// Forbidden symbols raise errSymbol (dna_string_n.hpp:87-93 exits 1).
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_rank.h"

// one workgroup iteration = one group of 256 rows = 8 blocks; one wave = 2 blocks
__global__ __launch_bounds__(256) void k_rank_build(const u8 *__restrict__ bwt, const u8 *__restrict__ qs, u64 n, u32 term,
                                                    RankBlock *__restrict__ blk, u32 *__restrict__ gcnt, u64 ngroups,
                                                    DevCounters *cnt)
{
    __shared__ u32 bc[8][6];          // per-block symbol counts of the group
    const u32 lane = bfq_lane(), w = threadIdx.x >> 6;
    for (u64 g = blockIdx.x; g < ngroups; g += gridDim.x) {
        u64 r = g * 256 + threadIdx.x;
        u32 code = 7;                               // rows past the end match no symbol
        u32 q = 0;
        if (r < n) {
            u8 ch = bwt[r];
            q = qs[r];
            code = (ch == (u8)term) ? 0u : bfq_base_code(ch);
            if (code == BFQ_CODE_INVALID) { atomicAdd(&cnt->errSymbol, 1ull); code = 4; }
            if (q & 0x80u) { atomicAdd(&cnt->errQual, 1ull); q &= 0x7Fu; }
        }
        u64 p0 = __ballot(code & 1u), p1 = __ballot(code & 2u), p2 = __ballot(code & 4u);
        u32 half = lane >> 5;                       // which of the wave's two blocks
        u32 h0 = (u32)(p0 >> (32 * half)), h1 = (u32)(p1 >> (32 * half)), h2 = (u32)(p2 >> (32 * half));
        RankBlock *B = &blk[g * 8 + w * 2 + half];
        B->q[lane & 31u] = (u8)q;
        u32 sub = lane & 31u;
        if (sub < 3) B->pl[sub] = sub == 0 ? h0 : (sub == 1 ? h1 : h2);
        if (sub < 6) bc[w * 2 + half][sub] = __popc(rank_match32(h0, h1, h2, sub));
        __syncthreads();
        // exclusive prefix of the 8 blocks inside the group, per symbol
        if (threadIdx.x < 48) {
            u32 b = threadIdx.x / 6, s = threadIdx.x % 6;
            u32 ex = 0;
            for (u32 k = 0; k < b; k++) ex += bc[k][s];
            RankBlock *D = &blk[g * 8 + b];
            if (s == 1) D->cnt[0] = ex; else if (s == 2) D->cnt[1] = ex; else if (s == 3) D->cnt[2] = ex;
            else if (s == 5) D->cnt[3] = ex; else if (s == 4) D->cntN = ex;
            if (b == 7) gcnt[(u64)s * ngroups + g] = ex + bc[7][s];
        }
        __syncthreads();
    }
}

// add (group absolute - super-block absolute) to the in-group prefixes; F array
__global__ __launch_bounds__(256) void k_rank_final(RankBlock *__restrict__ blk, const u64 *__restrict__ scanned,
                                                    u64 nblk, u64 ngroups, DevCounters *cnt, u64 *__restrict__ F)
{
    for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (u64)gridDim.x * blockDim.x) {
        u64 g = b >> 3;
        u64 sg = (g >> (BFQ_SUPER_SHIFT - BFQ_GROUP_SHIFT)) << (BFQ_SUPER_SHIFT - BFQ_GROUP_SHIFT);
        RankBlock *B = &blk[b];
        B->cnt[0] += (u32)(scanned[1 * ngroups + g] - scanned[1 * ngroups + sg]);
        B->cnt[1] += (u32)(scanned[2 * ngroups + g] - scanned[2 * ngroups + sg]);
        B->cnt[2] += (u32)(scanned[3 * ngroups + g] - scanned[3 * ngroups + sg]);
        B->cnt[3] += (u32)(scanned[5 * ngroups + g] - scanned[5 * ngroups + sg]);
        B->cntN += (u32)(scanned[4 * ngroups + g] - scanned[4 * ngroups + sg]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u64 acc = 0;
        for (int s = 0; s < 6; s++) { F[s] = acc; acc += cnt->tot[s]; }
    }
}

RankIndex bfq_rank_build(bfq_ctx *c, const u8 *bwt, const u8 *qs, u64 n, int term)
{
    RankIndex R;
    u64 ngroups = n / 256 + 1, nblk = ngroups * 8;
    RankBlock *blk = c->alloc<RankBlock>(nblk);
    u64 *scanned = c->alloc<u64>(6 * ngroups);
    u64 *F = c->alloc<u64>(8);
    size_t m = c->mark();
    u32 *gcnt = c->alloc<u32>(6 * ngroups);
    KLAUNCH(c, K_RANK_BUILD, 4.0 * (double)n, k_rank_build, bfq_grid(ngroups, 1), 256, bwt, qs, n, (u32)(term & 0xFF), blk,
            gcnt, ngroups, c->d_cnt);
    for (int s = 0; s < 6; s++) bfq_exscan_u32(c, gcnt + (u64)s * ngroups, scanned + (u64)s * ngroups, ngroups, &c->d_cnt->tot[s]);
    KLAUNCH(c, K_RANK_FINAL, 2.5 * (double)n, k_rank_final, bfq_grid(nblk, 256), 256, blk, (const u64 *)scanned, nblk,
            ngroups, c->d_cnt, F);
    c->release(m);
    R.blk = blk; R.scanned = scanned; R.F = F; R.n = n; R.ngroups = ngroups;
    return R;
}
