/*
 * bfq_oracle.c -- CPU restatement ("oracle") of BFQzip's hot path in plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see bfq_oracle.h).  Every function cites the
 * reference lines (paths relative to the BFQzip tree) whose behaviour it
 * restates.  Nothing here is tuned: it is the checker and the reported CPU
 * baseline ("port"), single-threaded like the reference.
 */
#define _GNU_SOURCE
#include "bfq_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- alphabet ---------------------------------------------------------- */
/* Order # < A < C < G < N < T: F array layout of dna_bwt_n.hpp:46-61
 * (F[A]=#TERM, then C, G, N, T), which is also plain ASCII order.          */
static inline int sym_code(uint8_t c, int term)
{
    if (c == (uint8_t)term) return 0;
    switch (c) {
    case 'A': return 1; case 'C': return 2; case 'G': return 3;
    case 'N': return 4; case 'T': return 5;
    default: return -1;
    }
}
static const uint8_t CODE2SYM[6] = {'#', 'A', 'C', 'G', 'N', 'T'};

/* bfq_int.cpp:106-110  DNA[]={A,C,G,T,N}, ORD: A0 C1 G2 T3 N4 */
static const uint8_t DNA5[5] = {'A', 'C', 'G', 'T', 'N'};
static inline int ord5(uint8_t c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2;
                 case 'T': return 3; case 'N': return 4; default: return 0; }
}

void orc_default_params(orc_params *p)
{
    p->K = 16; p->m = 2; p->v = '>'; p->f = 40; p->t = 20; p->term = '#';
    p->M = 2; p->B = 0; p->ext = 0;   /* bfq_int.cpp:70-90 */
}

/* bfq_int.cpp:307-319 illumina_8_level_binning (argument there is q-33) */
int orc_bin8(int ascii_q)
{
    int q = ascii_q - 33;
    if (q >= 40) q = 40; else if (q >= 35) q = 37; else if (q >= 30) q = 33;
    else if (q >= 25) q = 27; else if (q >= 20) q = 22; else if (q >= 10) q = 15;
    else if (q >= 2) q = 6;
    return q + 33;
}

/* ---- step 1: generalized suffix sort (contract of gsufsort --bwt --qs,
 *      call site BFQzip.py:178-189; source absent, see header) ------------- */
static const uint8_t *g_T;        /* text of codes, terminator = 0 */

static inline int suf_less(uint64_t p, uint64_t q)
{
    const uint8_t *a = g_T + p, *b = g_T + q;
    for (;;) {
        uint8_t x = *a++, y = *b++;
        if (x != y) return x < y;
        if (x == 0) return p < q;   /* #_i < #_j iff i < j */
    }
}
static int suf_cmp(const void *pa, const void *pb)
{
    uint64_t p = *(const uint64_t *)pa, q = *(const uint64_t *)pb;
    if (p == q) return 0;
    return suf_less(p, q) ? -1 : 1;
}

int orc_build_ebwt(const uint8_t *bases, const uint8_t *quals, const uint64_t *roff,
                   uint64_t N, int term, uint8_t *bwt, uint8_t *qs,
                   uint32_t *lcp, uint64_t *sa_out)
{
    uint64_t nb = roff[N], n = nb + N;
    uint8_t *T = (uint8_t *)malloc(n + 8);
    uint8_t *Q = (uint8_t *)malloc(n + 8);
    uint64_t *sa = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
    if (!T || !Q || !sa) { free(T); free(Q); free(sa); return -1; }
    memset(T + n, 0, 8);
    for (uint64_t i = 0; i < N; i++) {
        uint64_t b = roff[i], e = roff[i + 1], t0 = b + i;
        for (uint64_t k = b; k < e; k++) {
            int c = sym_code(bases[k], -1);   /* '#' never a base */
            if (c <= 0) { free(T); free(Q); free(sa); return -2; }
            T[t0 + (k - b)] = (uint8_t)c;
            Q[t0 + (k - b)] = quals[k];
        }
        T[t0 + (e - b)] = 0;
        Q[t0 + (e - b)] = '#';
    }
    /* bucket by the first 4 symbols (counting sort), then qsort each bucket */
    enum { BK = 6 * 6 * 6 * 6 };
    uint64_t *cnt = (uint64_t *)calloc(BK + 1, sizeof(uint64_t));
    uint32_t *bk = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint64_t p = 0; p < n; p++) {
        uint32_t k = 0; int dead = 0;
        for (int j = 0; j < 4; j++) {
            uint8_t c = dead ? 0 : T[p + j];
            if (c == 0) dead = 1;
            k = k * 6 + c;
        }
        bk[p] = k; cnt[k + 1]++;
    }
    for (int k = 0; k < BK; k++) cnt[k + 1] += cnt[k];
    {
        uint64_t *pos = (uint64_t *)malloc(sizeof(uint64_t) * BK);
        memcpy(pos, cnt, sizeof(uint64_t) * BK);
        for (uint64_t p = 0; p < n; p++) sa[pos[bk[p]]++] = p;
        free(pos);
    }
    g_T = T;
    for (int k = 0; k < BK; k++) {
        uint64_t lo = cnt[k], hi = cnt[k + 1];
        if (hi - lo > 1) qsort(sa + lo, hi - lo, sizeof(uint64_t), suf_cmp);
    }
    free(cnt); free(bk);
    for (uint64_t r = 0; r < n; r++) {
        uint64_t p = sa[r];
        /* bwt[r] = symbol preceding the suffix, TERM for a whole read */
        uint8_t c = (p == 0) ? 0 : T[p - 1];
        bwt[r] = c ? CODE2SYM[c] : (uint8_t)term;
        qs[r] = c ? Q[p - 1] : (uint8_t)'#';
        if (lcp) {
            uint32_t l = 0;
            if (r > 0) {
                const uint8_t *a = T + sa[r - 1], *b = T + p;
                while (*a && *a == *b) { a++; b++; l++; }  /* terminators never match */
            }
            lcp[r] = l;
        }
    }
    if (sa_out) memcpy(sa_out, sa, sizeof(uint64_t) * n);
    free(T); free(Q); free(sa);
    return 0;
}

/* ---- step 2: LCP flags ------------------------------------------------- */
/* Closed form of the suffix-tree navigation of bfq_int.cpp:139-181 (leaves /
 * minima) and :183-300 with include.hpp:888-925, identical to the streaming
 * 3-point rule of bfq_ext.cpp:377-392:
 *   thr[r] = r>=1 && LCP[r]>=K
 *   min[r] = 1<=r<=n-2 && LCP[r-1]>LCP[r] && LCP[r+1]>=LCP[r]
 * The cluster scan of bfq_int.cpp:685-711 only uses in = thr && !min.       */
void orc_flags(const uint32_t *lcp, uint64_t n, int K, uint8_t *in)
{
    for (uint64_t r = 0; r < n; r++) {
        int thr = (r >= 1) && ((int64_t)lcp[r] >= (int64_t)K);
        int mn = (r >= 1 && r + 2 <= n) && lcp[r - 1] > lcp[r] && lcp[r + 1] >= lcp[r];
        in[r] = (uint8_t)(thr && !mn);
    }
}

/* ---- rank support for LF (dna_bwt_n.hpp:80-101 LF, :46-61 F array) ------ */
typedef struct {
    const uint8_t *bwt; uint64_t n; int term;
    uint64_t *occ;      /* [n/64+1][6] */
    uint64_t F[6];
} lfidx;

static int lf_build(lfidx *x, const uint8_t *bwt, uint64_t n, int term)
{
    x->bwt = bwt; x->n = n; x->term = term;
    uint64_t nb = n / 64 + 1, c[6] = {0, 0, 0, 0, 0, 0};
    x->occ = (uint64_t *)malloc(sizeof(uint64_t) * 6 * nb);
    if (!x->occ) return -1;
    for (uint64_t i = 0; i < n; i++) {
        if ((i & 63) == 0) memcpy(x->occ + 6 * (i >> 6), c, sizeof(c));
        int s = sym_code(bwt[i], term);
        if (s < 0) { free(x->occ); return -2; }   /* dna_string_n.hpp:87-93 exit(1) */
        c[s]++;
    }
    if ((n & 63) == 0) memcpy(x->occ + 6 * (n >> 6), c, sizeof(c));
    x->F[0] = 0;
    for (int s = 1; s < 6; s++) x->F[s] = x->F[s - 1] + c[s - 1];
    return 0;
}
static inline uint64_t lf_map(const lfidx *x, uint64_t i)
{
    int s = sym_code(x->bwt[i], x->term);
    uint64_t r = x->occ[6 * (i >> 6) + s];
    for (uint64_t j = i & ~(uint64_t)63; j < i; j++) r += (sym_code(x->bwt[j], x->term) == s);
    return x->F[s] + r;
}

/* ---- step 3: clusters ---------------------------------------------------- */
typedef struct {
    const uint8_t *bwt; uint8_t *qual; uint8_t *modbit, *modsym;
    const orc_params *p; const lfidx *lf; orc_stats *st;
} clctx;

/* bfq_int.cpp:376-405 modBasesSmoothQS */
static void mod_smooth(clctx *c, uint64_t start, uint64_t end, uint8_t newSymb,
                       signed char newqs, const uint64_t *lowQS)
{
    uint8_t TERM = (uint8_t)c->p->term;
    for (uint64_t j = start; j <= end; j++) {
        uint8_t b = c->bwt[j];
        if (b == TERM) continue;
        if (b != newSymb && lowQS[ord5(b)] == 0) {
            c->modbit[j] = 1; c->modsym[j] = newSymb; c->st->modified++;
        } else if (b == newSymb) {
            c->qual[j] = (uint8_t)newqs; c->st->qs_smoothed++;
        } else if (newqs < (signed char)c->qual[j]) {
            c->qual[j] = (uint8_t)newqs; c->st->qs_smoothed++;
        }
    }
}

/* bfq_int.cpp:414-626 process_cluster(begin,i) with border=1 (:67) */
static void process_cluster(clctx *c, uint64_t begin, uint64_t i)
{
    const orc_params *P = c->p;
    uint8_t TERM = (uint8_t)P->term;
    uint64_t start = begin >= 1 ? begin - 1 : 0;
    uint64_t end = i > 1 ? i - 1 : 0;
    uint64_t size = end - start + 1;
    if (size < (uint64_t)P->m) return;                          /* :422 */

    uint64_t freqs[5] = {0}, lowQS[5] = {0}, base_num = 0;
    for (uint64_t j = start; j <= end; j++) {                    /* :437-449 */
        uint8_t b = c->bwt[j];
        if (b != TERM) {
            freqs[ord5(b)]++; base_num++;
            if ((signed char)c->qual[j] >= P->t + 33) lowQS[ord5(b)] = 1;
        }
    }
    c->st->num_clust++;
    if (base_num == 0) return;                                   /* :453 */
    c->st->bases_inside += base_num;

    signed char newqs;
    if (P->M == 1) {                                             /* :357-373 mean_error */
        double sum_err = 0; uint64_t num = 0;
        for (uint64_t j = start; j <= end; j++)
            if (c->bwt[j] != TERM) {
                num++;
                sum_err = sum_err + pow(10, -((double)(signed char)c->qual[j] - 33) / 10);
            }
        double avg_err = sum_err / num;
        int q = (int)round(-10 * log10(avg_err));
        if (P->ext) newqs = (signed char)(uint8_t)((uint8_t)q + 33);   /* bfq_ext.cpp:538-540 */
        else newqs = (signed char)(q + 33);
    } else if (P->M == 2) {
        newqs = (signed char)P->v;                               /* :467 */
    } else if (P->M == 3) {                                      /* :323-338 avg_qs */
        int sum = 0; uint64_t num = 0;
        for (uint64_t j = start; j <= end; j++)
            if (c->bwt[j] != TERM) { sum = sum + (int)(signed char)c->qual[j]; num++; }
        if (sum == 0) newqs = 0;
        else if (P->ext) newqs = (signed char)(uint8_t)round((float)sum / num); /* bfq_ext.cpp:496 */
        else newqs = (signed char)(int)((uint64_t)(int64_t)sum / num);
    } else {                                                     /* :342-353 max_qs */
        signed char mx = 0;
        for (uint64_t j = start; j <= end; j++)
            if (c->bwt[j] != TERM && (signed char)c->qual[j] > mx) mx = (signed char)c->qual[j];
        newqs = mx;
    }

    uint8_t Freq[5]; int nf = 0, nnn = 0;                        /* :480-499 */
    for (int s = 0; s < 5; s++)
        if (freqs[s] > 0) {
            nnn++;
            unsigned char perc = (unsigned char)((100 * freqs[s]) / base_num);
            if ((float)perc >= (float)P->f) Freq[nf++] = DNA5[s];
        }
    if (nnn == 1) c->st->num_clust_alleq++;
    if (nf >= 3) abort();                                        /* :505 assert */

    if (nf == 0) { c->st->num_clust_discarded++; return; }
    if (nf == 1) {
        if (Freq[0] == 'N') c->st->num_clust_discarded++;
        else mod_smooth(c, start, end, Freq[0], newqs, lowQS);
        return;
    }
    if (base_num < (uint64_t)P->m) { c->st->num_clust_discarded++; return; }   /* :520 */
    if (Freq[0] == 'N') { mod_smooth(c, start, end, Freq[1], newqs, lowQS); c->st->num_clust_mod++; return; }
    if (Freq[1] == 'N') { mod_smooth(c, start, end, Freq[0], newqs, lowQS); c->st->num_clust_mod++; return; }

    /* :542-565 predecessor symbols of the two frequent bases */
    uint8_t symbPrec[2] = {0, 0}; int fr[2][4] = {{0}}, tot[2] = {0, 0};
    for (uint64_t j = start; j <= end; j++) {
        int w = c->bwt[j] == Freq[0] ? 0 : (c->bwt[j] == Freq[1] ? 1 : -1);
        if (w < 0) continue;
        uint8_t ch = c->bwt[lf_map(c->lf, j)];
        if (ch != TERM && ch != 'N') { fr[w][ord5(ch)] = 1; symbPrec[w] = ch; }
    }
    for (int s = 0; s < 4; s++) { tot[0] += fr[0][s]; tot[1] += fr[1][s]; }
    if (tot[0] == 1 && tot[1] == 1 && symbPrec[0] != symbPrec[1]) {          /* :568 */
        c->st->num_clust_mod++;
        for (uint64_t j = start; j <= end; j++) {
            uint8_t b = c->bwt[j];
            if (b == TERM) continue;
            if (b != Freq[0] && b != Freq[1] && lowQS[ord5(b)] == 0) {
                uint8_t ch = c->bwt[lf_map(c->lf, j)];
                if (ch == symbPrec[0]) { c->modbit[j] = 1; c->modsym[j] = Freq[0]; c->st->modified++; }
                else if (ch == symbPrec[1]) { c->modbit[j] = 1; c->modsym[j] = Freq[1]; c->st->modified++; }
            } else if (b == Freq[0] || b == Freq[1]) {
                c->qual[j] = (uint8_t)newqs; c->st->qs_smoothed++;
            } else if (newqs < (signed char)c->qual[j]) {
                c->qual[j] = (uint8_t)newqs; c->st->qs_smoothed++;
            }
        }
    } else {
        c->st->num_clust_amb_discarded++;
    }
}

/* bfq_int.cpp:636-737 run(): the open/close scan over in = thr && !min */
static int smooth_with_lf(const uint8_t *bwt, uint8_t *qual, const uint8_t *in, uint64_t n,
                          const orc_params *p, const lfidx *lf, uint8_t *modbit,
                          uint8_t *modsym, orc_stats *st)
{
    clctx c = {bwt, qual, modbit, modsym, p, lf, st};
    memset(st, 0, sizeof(*st));
    memset(modbit, 0, n);
    memset(modsym, 0, n);
    uint64_t begin = 0; int open = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (in[i]) { if (!open) { open = 1; begin = i; } }
        else { if (open) process_cluster(&c, begin, i); open = 0; }
    }
    if (open) process_cluster(&c, begin, n);
    return 0;
}

int orc_smooth(const uint8_t *bwt, uint8_t *qual, const uint8_t *in, uint64_t n,
               const orc_params *p, uint8_t *modbit, uint8_t *modsym, orc_stats *st)
{
    lfidx lf;
    int rc = lf_build(&lf, bwt, n, p->term);
    if (rc) return rc;
    rc = smooth_with_lf(bwt, qual, in, n, p, &lf, modbit, modsym, st);
    free(lf.occ);
    return rc;
}

/* ---- step 4: inversion (bfq_int.cpp:748-819) ----------------------------- */
static int64_t invert_with_lf(const lfidx *lf, const uint8_t *qual, const uint8_t *modbit,
                              const uint8_t *modsym, int B, uint8_t *out_bases,
                              uint8_t *out_quals, uint64_t *out_roff)
{
    const uint8_t *bwt = lf->bwt; uint8_t TERM = (uint8_t)lf->term;
    uint64_t N = lf->F[1];            /* dna_bwt_n.hpp:364-366 number of strings = F[A] */
    uint64_t o = 0;
    for (uint64_t i = 0; i < N; i++) {
        out_roff[i] = o;
        uint64_t j = i, len = 0;
        while (bwt[j] != TERM) {      /* emitted back to front, reversed below */
            out_bases[o + len] = (modbit && modbit[j]) ? modsym[j] : bwt[j];
            out_quals[o + len] = B ? (uint8_t)orc_bin8((signed char)qual[j]) : qual[j];
            len++;
            j = lf_map(lf, j);
            if (o + len > lf->n) return -3;
        }
        for (uint64_t a = 0, b = len; a + 1 < b; a++) {
            b--;
            uint8_t t = out_bases[o + a]; out_bases[o + a] = out_bases[o + b]; out_bases[o + b] = t;
            t = out_quals[o + a]; out_quals[o + a] = out_quals[o + b]; out_quals[o + b] = t;
        }
        o += len;
    }
    out_roff[N] = o;
    return (int64_t)N;
}

int64_t orc_invert(const uint8_t *bwt, const uint8_t *qual, const uint8_t *modbit,
                   const uint8_t *modsym, uint64_t n, int term, int B,
                   uint8_t *out_bases, uint8_t *out_quals, uint64_t *out_roff)
{
    lfidx lf;
    int rc = lf_build(&lf, bwt, n, term);
    if (rc) return rc;
    int64_t N = invert_with_lf(&lf, qual, modbit, modsym, B, out_bases, out_quals, out_roff);
    free(lf.occ);
    return N;
}

/* LCP of a given eBWT.  The reference deduces it from the BWT alone by
 * Weiner-link navigation (bfq_int.cpp:183-300), seeing ONE terminator symbol: the
 * result is the LCP of the rows' suffixes with terminators never matching
 * (bfq_int.cpp:139-145), whatever the order of identical suffixes in the given eBWT.
 * Here: every row's suffix is located by the LF walks (row -> position in the decoded
 * reads), then adjacent rows are compared directly.                               */
int orc_lcp_from_bwt(const uint8_t *bwt, uint64_t n, int term, uint32_t *lcp)
{
    lfidx lf;
    int rc = lf_build(&lf, bwt, n, term);
    if (rc) return rc;
    uint64_t N = lf.F[1];
    uint8_t *T = (uint8_t *)malloc(n + 1);                 /* decoded reads, each followed by a 0 byte */
    uint64_t *pos = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint64_t *rows = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
    uint64_t o = 0, seen = 0;
    rc = 0;
    for (uint64_t i = 0; i < N && !rc; i++) {
        uint64_t j = i, len = 0;
        rows[0] = j;
        while (bwt[j] != (uint8_t)term) {                  /* step t visits the row of the suffix of length t */
            if (o + len + 1 >= n + 1) { rc = -3; break; }
            T[o + len] = bwt[j];                           /* back to front, reversed below */
            len++;
            j = lf_map(&lf, j);
            rows[len] = j;
        }
        if (rc) break;
        for (uint64_t a = 0, b = len; a + 1 < b; a++) { b--; uint8_t t = T[o + a]; T[o + a] = T[o + b]; T[o + b] = t; }
        T[o + len] = 0;
        for (uint64_t t = 0; t <= len; t++) pos[rows[t]] = o + len - t;
        seen += len + 1;
        o += len + 1;
    }
    free(lf.occ);
    if (!rc && seen != n) rc = -4;                         /* the walks do not cover the eBWT */
    if (!rc) {
        for (uint64_t r = 0; r < n; r++) {
            uint32_t l = 0;
            if (r > 0) {
                const uint8_t *a = T + pos[r - 1], *b = T + pos[r];
                while (*a && *a == *b) { a++; b++; l++; }  /* terminators never match */
            }
            lcp[r] = l;
        }
    }
    free(T); free(pos); free(rows);
    return rc;
}

int64_t orc_smooth_invert(const uint8_t *bwt, const uint8_t *bwtqs, const uint32_t *lcp_or_null,
                          uint64_t n, const orc_params *p,
                          uint8_t *out_bases, uint8_t *out_quals, uint64_t *out_roff,
                          orc_stats *st)
{
    uint32_t *lcp_own = NULL;
    const uint32_t *lcp = lcp_or_null;
    if (!lcp) {
        lcp_own = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
        int rc = orc_lcp_from_bwt(bwt, n, p->term, lcp_own);
        if (rc) { free(lcp_own); return rc; }
        lcp = lcp_own;
    }
    uint8_t *in = (uint8_t *)malloc(n + 1), *qual = (uint8_t *)malloc(n + 1);
    uint8_t *modbit = (uint8_t *)malloc(n + 1), *modsym = (uint8_t *)malloc(n + 1);
    memcpy(qual, bwtqs, n);
    orc_flags(lcp, n, p->K, in);
    lfidx lf;
    int64_t N = lf_build(&lf, bwt, n, p->term);
    if (N == 0) {
        smooth_with_lf(bwt, qual, in, n, p, &lf, modbit, modsym, st);
        N = invert_with_lf(&lf, qual, modbit, modsym, p->B, out_bases, out_quals, out_roff);
        free(lf.occ);
    }
    free(in); free(qual); free(modbit); free(modsym); free(lcp_own);
    return N;
}

int orc_run_reads(const uint8_t *bases, const uint8_t *quals, const uint64_t *roff, uint64_t N,
                  const orc_params *p, uint8_t *out_bases, uint8_t *out_quals, orc_stats *st)
{
    uint64_t n = roff[N] + N;
    uint8_t *bwt = (uint8_t *)malloc(n + 1), *qs = (uint8_t *)malloc(n + 1);
    uint32_t *lcp = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint64_t *oroff = (uint64_t *)malloc(sizeof(uint64_t) * (N + 1));
    int rc = orc_build_ebwt(bases, quals, roff, N, p->term, bwt, qs, lcp, NULL);
    if (rc == 0) {
        int64_t r = orc_smooth_invert(bwt, qs, lcp, n, p, out_bases, out_quals, oroff, st);
        if (r != (int64_t)N) rc = -6;
        else for (uint64_t i = 0; i <= N; i++) if (oroff[i] != roff[i]) { rc = -7; break; }
    }
    free(bwt); free(qs); free(lcp); free(oroff);
    return rc;
}
