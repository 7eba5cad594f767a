// bfq_synth.h -- seeded, counter-based synthetic short reads (host and device).
//
// The reference ships no generator (randomFASTQ.py only shuffles an existing
// file, randomFASTQ.py:52-102), so the benchmark workload is defined here:
// a random genome of length G = N*Lavg/coverage, two haplotypes (SNP about
// every snp_every bases plus pairs of adjacent SNPs about every dsnp_every),
// reads drawn uniformly from either haplotype and (optionally) either strand,
// substitution errors with low quality, a few 'N' calls with quality '#'.
// Every byte is a pure function of (seed, read index, offset): any block of
// reads can be generated independently on any GPU or on the host (bfq_synth.first /
// .collection: the reads of the spec are reads [first, first+N) of the collection;
// the functions below take the read's index in the collection).
#pragma once
#include "bfq_common.h"
#include "../../include/bfqzip_hip.h"

BFQ_HD u64 bfq_synth_genome_len(const bfq_synth *s)
{
    u64 lavg = ((u64)s->Lmin + s->Lmax) / 2;
    u64 cov = s->coverage ? s->coverage : 1;
    u64 G = (s->collection ? s->collection : s->N) * lavg / cov;
    u64 minG = (u64)s->Lmax * 2 + 16;
    return G < minG ? minG : G;
}
BFQ_HD u32 bfq_synth_len(const bfq_synth *s, u64 i)
{
    if (s->Lmin >= s->Lmax) return s->Lmin;
    return s->Lmin + (u32)(bfq_hash2(s->seed, 7, i) % (u64)(s->Lmax - s->Lmin + 1));
}
// haplotype base (0..3 = A,C,G,T) at genome position g
BFQ_HD u32 bfq_synth_hap_base(const bfq_synth *s, u64 g, u32 hap)
{
    u32 b = (u32)(bfq_hash2(s->seed, 1, g) & 3);
    if (hap) {
        bool snp = s->snp_every && (bfq_hash2(s->seed, 2, g) % s->snp_every) == 0;
        if (s->dsnp_every) {
            snp = snp || (bfq_hash2(s->seed, 3, g) % s->dsnp_every) == 0;
            snp = snp || (g > 0 && (bfq_hash2(s->seed, 3, g - 1) % s->dsnp_every) == 0);
        }
        if (snp) b = (b + 1 + (u32)(bfq_hash2(s->seed, 4, g) % 3)) & 3;
    }
    return b;
}
// base and quality (ASCII) of read i (length len) at offset k
BFQ_HD void bfq_synth_base(const bfq_synth *s, u64 i, u32 len, u32 k, u8 *base, u8 *qual)
{
    const u8 ACGT[4] = {'A', 'C', 'G', 'T'};
    const u8 QS[8] = {12, 18, 25, 30, 35, 38, 40, 41};
    u64 G = bfq_synth_genome_len(s);
    u64 h = bfq_hash2(s->seed, 5, i);
    u64 start = h % (G - len + 1);
    u32 hap = (u32)(h >> 61) & 1, rev = s->both_strands ? (u32)(h >> 62) & 1 : 0;
    u64 g = rev ? start + (len - 1 - k) : start + k;
    u32 b = bfq_synth_hap_base(s, g, hap);
    if (rev) b = 3 - b;
    u64 e = bfq_hash2(s->seed, 6, (i << 16) ^ (u64)k ^ (i >> 48 << 60));
    u32 r = (u32)(e % 1000000ull);
    u32 q;
    if (r < s->n_ppm) { *base = 'N'; *qual = '#'; return; }
    if (r < s->n_ppm + s->err_ppm) {
        b = (b + 1 + (u32)((e >> 24) % 3)) & 3;
        q = 2 + (u32)((e >> 32) % 24);
    } else {
        q = QS[(e >> 40) & 7];
    }
    *base = ACGT[b];
    *qual = (u8)(33 + q);
}
