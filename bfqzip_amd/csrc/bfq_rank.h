// bfq_rank.h -- the GPU-resident LF table that the cluster step and the inversion query.
//
// The reference answers LF(j) = F[c] + rank_c(j) on demand from its succinct BWT
// (external/bwt2lcp/dna_string_n.hpp:112-185 operator[] / parallel_rank / rank,
// dna_bwt_n.hpp:80-101 LF) plus a QUAL[] byte and a rankbv lookup for replaced bases
// (external/rankbv/rankbv.cpp:115-135, bfq_int.cpp:782).  On MI355X the rank queries of
// ALL rows are answered once, in row order, by k_lf_build (per-lane symbol counters packed in a
// u64 + wave scan + scanned group counters), and tabulated with everything else a walk needs:
//
//   one u64 per eBWT row:  bits  0..39  LF(row)           (0 for terminator rows)
//                          bits 40..42  symbol code       (# 0, A 1, C 2, G 3, N 4, T 5)
//                          bit  43      base replaced     (reference: rankbv bit)
//                          bits 44..46  replacement code  (reference: BWT_MOD entry)
//                          bits 48..55  quality byte      (reference: QUAL[row], edited in place)
//
// so one LF step of the inversion is ONE 8-byte access (one HBM sector) instead of a
// 64-byte block + QUAL byte + rankbv word, and needs no popcounts.  8 bytes per row:
// 36 GB at 30 M x 150 bp, affordable in 288 GB of HBM and free after step 1.
#pragma once
#include "bfq_internal.h"

#define LFQ_POS_MASK ((1ull << 40) - 1ull)
__device__ __forceinline__ u64 lfq_next(u64 x) { return x & LFQ_POS_MASK; }
__device__ __forceinline__ u32 lfq_code(u64 x) { return (u32)(x >> 40) & 7u; }
__device__ __forceinline__ bool lfq_replaced(u64 x) { return (x >> 43) & 1ull; }
__device__ __forceinline__ u32 lfq_repl(u64 x) { return (u32)(x >> 44) & 7u; }
__device__ __forceinline__ u32 lfq_qual(u64 x) { return (u32)(x >> 48) & 0xFFu; }
// byte stores into an entry: only the thread that owns the row's cluster writes them
__device__ __forceinline__ void lfq_set_qual(u64 *lfq, u64 j, u32 q) { ((u8 *)(lfq + j))[6] = (u8)q; }
__device__ __forceinline__ void lfq_set_repl(u64 *lfq, u64 j, u32 code, u32 repl)
{
    ((u8 *)(lfq + j))[5] = (u8)(code | 0x08u | (repl << 4));
}
