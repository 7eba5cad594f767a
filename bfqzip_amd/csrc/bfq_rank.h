// bfq_rank.h -- device-side queries on the GPU-resident rank structure of the eBWT.
//
// Replaces the reference's succinct BWT (external/bwt2lcp/dna_string_n.hpp:112-185
// operator[] / parallel_rank / rank, dna_bwt_n.hpp:80-101 LF) and its rankbv
// (external/rankbv/rankbv.cpp:115-135 access / rank1): same answers, own layout.
//
// One 64-byte block per 32 rows holds EVERYTHING one LF step of the inversion needs:
//   u32 cnt[4] : A,C,G,T before the block, relative to the enclosing super block
//   u32 pl[3]  : bit planes of the symbol code (# 0, A 1, C 2, G 3, N 4, T 5)
//   u32 cntN   : N before the block (relative)
//   u8  q[32]  : quality of the row; bit 7 set = "base replaced" (symbol in modsym[])
// so a step is ONE 64-byte HBM access (the reference touches a 64-B BWT block, the
// QUAL byte and the rankbv word).  Super blocks = 2^21 rows; their absolute
// counters are the scanned per-256-row group totals (L2-resident table).
#pragma once
#include "bfq_internal.h"

#define BFQ_SUPER_SHIFT 21                       // rows per super block
#define BFQ_GROUP_SHIFT 8                        // rows per counted group (k_rank_build workgroup)

typedef RankIndex RankDev;

__device__ __forceinline__ u32 rank_match32(u32 p0, u32 p1, u32 p2, u32 code)
{
    u32 m0 = (code & 1u) ? p0 : ~p0;
    u32 m1 = (code & 2u) ? p1 : ~p1;
    u32 m2 = (code & 4u) ? p2 : ~p2;
    return m0 & m1 & m2;
}

// index of a base code 1..5 into the scanned table rows: # 0, A 1, C 2, G 3, N 4, T 5 (same as the code)
__device__ __forceinline__ u64 rank_super(const RankDev &R, u64 j, u32 code)
{
    u64 g = (j >> BFQ_SUPER_SHIFT) << (BFQ_SUPER_SHIFT - BFQ_GROUP_SHIFT);   // first group of the super block
    return R.scanned[(u64)code * R.ngroups + g];
}

struct RankHdr { u32 cnt[4]; u32 pl[3]; u32 cntN; };   // first 32 bytes of a block

__device__ __forceinline__ RankHdr rank_load_hdr(const RankDev &R, u64 j)
{
    const uint4 *p = (const uint4 *)&R.blk[j >> 5];
    uint4 a = p[0], b = p[1];
    RankHdr h;
    h.cnt[0] = a.x; h.cnt[1] = a.y; h.cnt[2] = a.z; h.cnt[3] = a.w;
    h.pl[0] = b.x; h.pl[1] = b.y; h.pl[2] = b.z; h.cntN = b.w;
    return h;
}
__device__ __forceinline__ u32 rank_hdr_code(const RankHdr &h, u64 j)
{
    u32 bit = (u32)j & 31u;
    return ((h.pl[0] >> bit) & 1u) | (((h.pl[1] >> bit) & 1u) << 1) | (((h.pl[2] >> bit) & 1u) << 2);
}
// LF(j) for a row holding base `code` (1..5): F[code] + #code in rows [0,j)
__device__ __forceinline__ u64 rank_hdr_lf(const RankDev &R, const RankHdr &h, u64 j, u32 code)
{
    u32 bit = (u32)j & 31u;
    u32 rel = (code == 4u) ? h.cntN : h.cnt[code == 5u ? 3u : code - 1u];
    u32 in = __popc(rank_match32(h.pl[0], h.pl[1], h.pl[2], code) & ((1u << bit) - 1u));
    return R.F[code] + rank_super(R, j, code) + (u64)rel + (u64)in;
}

__device__ __forceinline__ u32 rank_code_at(const RankDev &R, u64 j)
{
    const RankBlock &B = R.blk[j >> 5];
    u32 bit = (u32)j & 31u;
    return ((B.pl[0] >> bit) & 1u) | (((B.pl[1] >> bit) & 1u) << 1) | (((B.pl[2] >> bit) & 1u) << 2);
}
__device__ __forceinline__ u64 rank_lf(const RankDev &R, u64 j, u32 code)
{
    RankHdr h = rank_load_hdr(R, j);
    return rank_hdr_lf(R, h, j, code);
}
