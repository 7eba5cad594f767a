/*
 * bfq_oracle.h -- CPU restatement ("oracle") of BFQzip's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may call it, and only as the checker / reported baseline.
 *
 * Parity status: PINNED for steps 2-4 (bfq_int) against the reference itself,
 * compiled from /root/reference by oracle/Makefile into oracle/_ref/ and
 * compared byte-for-byte (tests/golden/ holds the resulting vectors, made by
 * tests/golden/make_golden.py).  Step 1 (gsufsort / eGap) is an un-vendored,
 * un-pinned submodule (felipelouza/gsufsort, felipelouza/egap, empty dirs in
 * the reference tree): its output contract is restated from the reference's
 * consumers (bfq_int.cpp:775-791 starts read i at BWT row i; dna_bwt_n.hpp:46-61
 * lays F out as # A C G N T) and is pinned only through the round trip
 * "oracle eBWT -> reference bfq_int -k 10000 -> original reads".
 */
#ifndef BFQ_ORACLE_H
#define BFQ_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int K;      /* -k  minimum LCP inside clusters          (bfq_int.cpp:70,931)  */
    int m;      /* -m  minimum cluster length                (bfq_int.cpp:74,932)  */
    int v;      /* -v  replacement quality, ASCII code (M=2) (bfq_int.cpp:78,933)  */
    int f;      /* -f  frequent-symbol percentage            (bfq_int.cpp:86,935)  */
    int t;      /* -t  trusted-quality threshold (phred)     (bfq_int.cpp:82,934)  */
    int term;   /* -s  terminator byte                       (bfq_int.cpp:90,913)  */
    int M;      /* compile-time -DM=0..3 of the reference    (bfq_int.cpp:462-473) */
    int B;      /* compile-time -DB=0/1 of the reference     (bfq_int.cpp:784-786) */
    int ext;    /* 1: bfq_ext rounding of M=3 (bfq_ext.cpp:496), else bfq_int      */
} orc_params;

typedef struct {
    uint64_t num_clust, num_clust_discarded, num_clust_amb_discarded,
             num_clust_mod, num_clust_alleq, bases_inside, qs_smoothed, modified;
} orc_stats;

void orc_default_params(orc_params *p);

/* Step 1: eBWT + permuted qualities + LCP (+SA) of a read collection.
 * bases/quals: concatenated reads (no separators); roff[N+1] offsets.
 * n = roff[N] + N rows.  lcp (uint32, n entries) and sa (uint64, text
 * positions, n entries) may be NULL. Returns 0 on success. */
int orc_build_ebwt(const uint8_t *bases, const uint8_t *quals, const uint64_t *roff,
                   uint64_t N, int term, uint8_t *bwt, uint8_t *qs,
                   uint32_t *lcp, uint64_t *sa);

/* in(r) = thr(r) && !min(r)  (closed form of bfq_int.cpp:139-181,183-300). */
void orc_flags(const uint32_t *lcp, uint64_t n, int K, uint8_t *in);

/* Step 3 (bfq_int.cpp:636-737 run, 414-626 process_cluster): edits qual in
 * place, sets modbit[r] (0/1) and modsym[r] (replacement symbol). */
int orc_smooth(const uint8_t *bwt, uint8_t *qual, const uint8_t *in, uint64_t n,
               const orc_params *p, uint8_t *modbit, uint8_t *modsym, orc_stats *st);

/* Step 4 (bfq_int.cpp:748-819 invert): LF walk from row i for read i.
 * out_roff must hold N+1 entries where N = number of terminator rows.
 * out_bases/out_quals must hold n-N bytes. Returns N (or <0 on error). */
int64_t orc_invert(const uint8_t *bwt, const uint8_t *qual, const uint8_t *modbit,
                   const uint8_t *modsym, uint64_t n, int term, int B,
                   uint8_t *out_bases, uint8_t *out_quals, uint64_t *out_roff);

/* LCP of a given eBWT (what bfq_int deduces by suffix-tree navigation,
 * bfq_int.cpp:183-300): restated as invert -> rebuild -> direct LCP. */
int orc_lcp_from_bwt(const uint8_t *bwt, uint64_t n, int term, uint32_t *lcp);

/* bfq_int / bfq_ext as a function: eBWT+QS(+LCP or NULL) -> smoothed reads. */
int64_t orc_smooth_invert(const uint8_t *bwt, const uint8_t *bwtqs, const uint32_t *lcp_or_null,
                          uint64_t n, const orc_params *p,
                          uint8_t *out_bases, uint8_t *out_quals, uint64_t *out_roff,
                          orc_stats *st);

/* The whole path: reads in -> smoothed reads out (same offsets). */
int orc_run_reads(const uint8_t *bases, const uint8_t *quals, const uint64_t *roff, uint64_t N,
                  const orc_params *p, uint8_t *out_bases, uint8_t *out_quals, orc_stats *st);

/* Illumina 8-level binning, ASCII in -> ASCII out (bfq_int.cpp:307-319). */
int orc_bin8(int ascii_q);

#ifdef __cplusplus
}
#endif
#endif
