// bfq_rankblk.h -- the 64-byte rank block (1 byte per eBWT row) and the queries answered from it: what the interval
// refinement of k_bfs.hip navigates by, and what the compact steps 2-4 (k_compact.hip: a given eBWT under a workspace cap)
// use INSTEAD of the 8-byte LF table entry -- the counterpart of the reference's dna_string_n block (117 symbols x 3 bit
// planes + in-block counters, dna_string_n.hpp:112-185,367-406) and of dna_bwt_n::LF (dna_bwt_n.hpp:80-101).
//   block b covers rows [64 b, 64 b + 64): u64 count of A, C, G, N, T before the block, then the three bit planes of the
//   rows' symbol codes (# 0, A 1, C 2, G 3, N 4, T 5).  Built by k_rankblocks (k_bfs.hip).
#pragma once
#include "bfq_internal.h"

struct RankBlk { u64 w[8]; };
// random blocks are read past the L1 (nontemporal: a block is used once per level / step)
__device__ __forceinline__ RankBlk load_blk(const u64 *__restrict__ rank, u64 blk)
{
    RankBlk b;
    const u64 *p = rank + (blk << 3);
#pragma unroll
    for (int k = 0; k < 8; k++) b.w[k] = __builtin_nontemporal_load(p + k);
    return b;
}
// occurrences of the codes 1..5 in rows [0, p), p inside (or at the end of) block b
__device__ __forceinline__ void occ5(const RankBlk &b, u64 p, u64 *o)
{
    const u32 k = (u32)p & 63u;
    const u64 lo = k ? (~0ull >> (64u - k)) : 0ull;
    const u64 p0 = b.w[5] & lo, p1 = b.w[6] & lo, p2 = b.w[7] & lo, n0 = ~b.w[5] & lo, n1 = ~b.w[6] & lo, n2 = ~b.w[7] & lo;
    o[0] = b.w[0] + (u64)__popcll(p0 & n1 & n2);      // A 001
    o[1] = b.w[1] + (u64)__popcll(n0 & p1 & n2);      // C 010
    o[2] = b.w[2] + (u64)__popcll(p0 & p1 & n2);      // G 011
    o[3] = b.w[3] + (u64)__popcll(n0 & n1 & p2);      // N 100
    o[4] = b.w[4] + (u64)__popcll(p0 & n1 & p2);      // T 101
}
// symbol code of row j (inside block b) and, for a base, the occurrences of that code in rows [0, j): one LF step
__device__ __forceinline__ u32 blk_code(const RankBlk &b, u64 j)
{
    const u32 k = (u32)j & 63u;
    return (u32)((b.w[5] >> k) & 1ull) | ((u32)((b.w[6] >> k) & 1ull) << 1) | ((u32)((b.w[7] >> k) & 1ull) << 2);
}
__device__ __forceinline__ u64 blk_occ_of(const RankBlk &b, u64 j, u32 code)   // code 1..5
{
    const u32 k = (u32)j & 63u;
    const u64 lo = k ? (~0ull >> (64u - k)) : 0ull;
    const u64 m = ((code & 1u) ? b.w[5] : ~b.w[5]) & ((code & 2u) ? b.w[6] : ~b.w[6]) & ((code & 4u) ? b.w[7] : ~b.w[7]) & lo;
    return b.w[code - 1] + (u64)__popcll(m);
}
