// bfq_rank.h -- device-side queries on the GPU-resident rank structure of the eBWT.
//
// Replaces the reference's succinct BWT (external/bwt2lcp/dna_string_n.hpp:112-185
// operator[] / parallel_rank / rank, dna_bwt_n.hpp:80-101 LF): same answers, own
// layout.  256 rows per 128-byte block: four 64-bit counters (A,C,G,T before the
// block) + 4 groups x 3 bit planes of the symbol code (# 0, A 1, C 2, G 3, N 4, T 5);
// the N counters live in a side array that is only touched when the row holds an N.
// One LF step = one 128-byte block read + <= 4 popcounts.
#pragma once
#include "bfq_internal.h"

typedef RankIndex RankDev;

__device__ __forceinline__ u64 rank_match(const u64 *pl, u32 code)
{
    u64 m0 = (code & 1u) ? pl[0] : ~pl[0];
    u64 m1 = (code & 2u) ? pl[1] : ~pl[1];
    u64 m2 = (code & 4u) ? pl[2] : ~pl[2];
    return m0 & m1 & m2;
}

// symbol code of row j
__device__ __forceinline__ u32 rank_code_at(const RankDev &R, u64 j)
{
    const RankBlock &B = R.blk[j >> 8];
    u32 g = (u32)(j >> 6) & 3u, bit = (u32)j & 63u;
    return (u32)((B.pl[g][0] >> bit) & 1ull) | ((u32)((B.pl[g][1] >> bit) & 1ull) << 1) |
           ((u32)((B.pl[g][2] >> bit) & 1ull) << 2);
}

// LF(j) for a row holding base `code` (1..5): F[code] + #code in rows [0,j)
__device__ __forceinline__ u64 rank_lf(const RankDev &R, u64 j, u32 code)
{
    const RankBlock &B = R.blk[j >> 8];
    u32 g = (u32)(j >> 6) & 3u, bit = (u32)j & 63u;
    u64 r;
    if (code == 4u) r = R.cntN[j >> 8];
    else r = B.cnt[code == 5u ? 3u : code - 1u];
    for (u32 q = 0; q < g; q++) r += (u64)__popcll(rank_match(B.pl[q], code));
    r += (u64)__popcll(rank_match(B.pl[g], code) & ((1ull << bit) - 1ull));
    return R.F[code] + r;
}

// code of row j and, if it is a base, LF(j) -- one block read for both
__device__ __forceinline__ u32 rank_step(const RankDev &R, u64 j, u64 *next)
{
    u32 code = rank_code_at(R, j);
    if (code) *next = rank_lf(R, j, code);
    return code;
}
