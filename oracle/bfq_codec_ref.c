/* bfq_codec_ref.c -- CPU statement of the stream codec (the entropy-coding back end that takes the
 * place of step 5, BFQzip.py:253-275: `7z a -mm=PPMd OUT.fq.dna.7z OUT.fq.dna`, `bsc e OUT.fq.qs OUT.fq.qs.bsc -T`).
 * TEST INFRASTRUCTURE ONLY: tests/ compare the GPU codec (bfqzip_amd/csrc/k_codec.hip) with this file byte for byte.
 *
 * PARITY UNPINNED against the reference's tools: 7z and libbsc are external (libbsc an empty submodule, 7z a system
 * binary); neither their sources nor their outputs are in /root/reference.  The container below is this project's own,
 * chosen for the GPU: a static order-k model per stream + range-ANS (Duda's rANS, byte-wise renormalisation), every
 * segment of 8192 symbols coded on its own so that encoder and decoder run one segment per lane.
 *
 * Line-structured streams (read names) first go through a LINE-DELTA transform when that shrinks them to 3/4 or less:
 *   char magic[8] = "BFQLINE1" | u64 raw_len | u32 R (= 256) | u32 0 | u64 number of lines | one BFQRANS2 container of the
 *   transformed bytes: per line (bytes up to and including its '\n'; the stream must end with one) a record
 *   byte(p + (p >= 10)) + line[p:], p = 0 for every R-th line, else the length of the prefix shared with the line before,
 *   capped at 254 and at the line's length - 1 (the record keeps the line's '\n', and no other byte of it is a '\n').
 * A file may hold several containers of any kind back to back (one per block of a sharded run).
 *
 * READ-ORDER DNA (OUT.fq.dna: lines of A C G T N, every one ending with '\n') takes the BFQDNAC1 container below when the
 * stream is nothing else: the redundancy of a 30x collection is between reads that cover the same stretch of the genome, 16
 * and more symbols of context away from what an order-7 table sees (2.06 bits per base); streams of 64 KiB and more only,
 * lines of 15 bases or more on average and none beyond 65535; and when the result is above 1.6 bits per base (little coverage:
 * nothing to learn from) and the plain BFQRANS2 container of the same bytes is smaller, that one is the output.  What PPMd / bsc do with an adaptive
 * model, symbol by symbol, is done here block by block so that both directions stay parallel:
 *   bases   = the lines without their '\n' (symbols A 0, C 1, G 2, T 3, N 4); lens[i] = length of line i (u32)
 *   segment g = the reads whose first base has an index in [g S, (g+1) S) of `bases` (S = 1024; a segment may be empty)
 *   block b   = segments [B_b, B_b+1), B_0 = 0, B_b+1 - B_b = min(16 << min(b, 12), clamp(nseg / 64, 256, 65536)):
 *               small blocks first (the table learns fastest at the start), 64 or more in all
 *   table     = 2^H rows, a row = one 64-bit word holding five 12-bit counters (symbol s at bits 12 s .. 12 s + 11), all
 *               zero at the start; H = clamp(ceil(log2(nbases)), 12, 32): the contexts that sequencing errors make need room
 *               too (30 M x 150 bp at 30x, bits per base: 1.08 with 2^30 rows, 0.86 with 2^31, 0.72 with 2^32).
 *               count_s = min(field_s, CAP), CAP = 4080 / W, W = 64.
 * A block is coded against the table as it stands BEFORE the block (frozen), then the table takes the block's symbols in:
 * encoder and decoder see the same table, nothing of it is stored, and inside a block every segment is independent.
 *   context of base j of a read: the last kk = min(j, K) pushed symbols p[] of the SAME read, as the number
 *     (K = clamp(ceil(log4(nbases)) + DNAC_KPLUS, 10, 20): long enough to be unique in a genome the collection covers)
 *     ctx = sum p[j-kk+t] << 3 (kk-1-t); row = mix(ctx * 32 + kk) >> (64 - H), mix = the splitmix64 finaliser
 *   model of a row, c_s = count_s: v_s = c_s * W + prior_s (prior 3 3 3 3 1), T = sum v_s,
 *     f_s = v_s * (2^12 - 5) / T + 1, the remainder up to 2^12 goes to the largest f (lowest symbol among equals)
 *   pushed symbol (what later contexts of the read see): the symbol itself, except that a base the row calls an error is
 *     replaced by the row's favourite -- j >= K, m = argmax c[0..3] (lowest among equals), c_m >= 3, c_actual == 0 and
 *     sum(c) - c_m <= c_m / 8: p[j] = m.  A sequencing error then costs one expensive symbol instead of K novel contexts.
 *   a base the frozen row knows well already (count of the actual symbol >= TSKIP = 8) leaves the table alone: at 30x that is
 *     most bases of the later blocks, and with them most of the update's memory traffic, for 0.2 % of the size
 *   update after the block, for every other base: the row's WORD += 1 << 12 s (mod 2^64: a field that passes 4095 carries into
 *     its neighbour -- only contexts seen thousands of times per block, and any order of the additions gives the same word):
 *     row(ctx, kk) += 1 << 12 (actual symbol);  and, when j >= K and p[j-K..j] are all bases (< 4), the other strand:
 *     ctx' = sum (3 - p[j-t]) << 3 (K-1-t), t = 0..K-1;  row(ctx', K) += 1 << 12 (3 - p[j-K])
 *   rANS exactly as below, one stream per non-empty segment (reads in order, symbols coded last to first).
 * Container:  "BFQDNAC1" | u64 raw_len | u64 nreads | u64 nbases | u32 K, H, S, nseg, scale_bits, W + (TSKIP << 8) | u64 checksum of the
 *   raw bytes | u64 L | a BFQRANS2 container (L bytes) of lens[] as little-endian u32 | u32 seg_bytes[nseg] | payload
 *
 * Container (little endian):
 *   char  magic[8] = "BFQRANS2"
 *   u64   raw_len
 *   u32   seg_syms (8192; halved down to 1024 while raw_len / seg_syms < 65536, so that short streams still fill the GPU),
 *         nseg = ceil(raw_len / seg_syms), A (distinct byte values), k (context order), scale_bits (12)
 *   u64   checksum of the raw bytes (below): the decoder recomputes it from what it decoded and refuses a mismatch --
 *         7z and bsc, whose place this takes, verify a CRC too; without it payload damage that still parses decodes to
 *         plausible wrong reads
 *   u8    alphabet[256]          byte value of symbol 0..A-1, ascending; the rest 0
 *   u16   dflt[A]                order-0 row: the model of every context without a row of its own
 *   u8    used[ceil(A^k / 8)]    bit c: context c has a row (it occurs in the sampled segments)
 *   u16   freq[used contexts][A] in ascending context order, every row sums to 2^scale_bits
 *   u32   seg_bytes[nseg]
 *   u8    payload[]              the segments' rANS streams back to back
 * Context of a symbol = the k symbols before it INSIDE its segment (missing ones count as symbol 0), read as a base-A
 * number, oldest symbol most significant.  k: counts are taken at the largest order kmax with A^(kmax+1) <= min(2^22,
 * max(4096, sampled_len / 16)), at most 8; the container is made with the order <= kmax whose estimated size (payload + table,
 * choose_order() below) is the smallest.
 * The model is counted on a SAMPLE: segments g with g % S == 0, S = clamp(raw_len / 2^24, 1, 64) (k is chosen for the
 * sampled length raw_len / S) -- a histogram of every symbol
 * of a 4.5 GB stream into a table of millions of bins is the one step that does not parallelise cheaply.  Every symbol of the
 * alphabet keeps a non-zero share in every row, so whatever the unsampled segments hold can be coded:
 * f[s] = max(1, floor(count[s] * 2^scale / total)) for ALL s; a surplus over 2^scale is taken from the largest entries
 * (lowest symbol first among equals, never below 1), a deficit goes to the largest entry.
 * rANS: state x in [2^23, 2^31), symbols coded last to first, bytes emitted low byte first and stored backwards, so the
 * decoder reads forwards: x = le32, then per symbol slot = x & (2^scale - 1), s = symbol with cum[s] <= slot < cum[s] + f[s],
 * x = f[s] * (x >> scale) + slot - cum[s], while x < 2^23: x = x << 8 | next byte.
 * Checksum: the raw bytes as little-endian 64-bit words w_0, w_1, ... (the last one zero-padded),
 *   mix(raw_len ^ sum_j mix(w_j + (j + 1) * 0x9E3779B97F4A7C15))  mod 2^64,  mix = the splitmix64 finaliser:
 * a sum of position-keyed terms, so any order of evaluation (one word per GPU lane) gives the same value.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BQC_SEG_MAX 8192u
#define BQC_SCALE 12u
#define BQC_L (1u << 23)

static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void put64(uint8_t *p, uint64_t v) { put32(p, (uint32_t)v); put32(p + 4, (uint32_t)(v >> 32)); }
static uint32_t get32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t get64(const uint8_t *p) { return (uint64_t)get32(p) | ((uint64_t)get32(p + 4) << 32); }

#define BQC_HDR 44u                                 /* magic, raw_len, five u32, checksum */
static uint64_t mix64(uint64_t z) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z; }
uint64_t orc_codec_checksum(const uint8_t *in, uint64_t n)
{
    uint64_t sum = 0;
    for (uint64_t j = 0; 8 * j < n; j++) {
        uint64_t w = 0;
        const uint64_t m = n - 8 * j < 8 ? n - 8 * j : 8;
        for (uint64_t b = 0; b < m; b++) w |= (uint64_t)in[8 * j + b] << (8 * b);
        sum += mix64(w + (j + 1) * 0x9E3779B97F4A7C15ull);
    }
    return mix64(n ^ sum);
}

static uint32_t choose_k(uint32_t A, uint64_t n)
{
    uint64_t limit = n >> 4;
    if (limit < 4096) limit = 4096;
    if (limit > (1u << 22)) limit = 1u << 22;
    uint32_t k = 0;
    uint64_t p = (uint64_t)A * A;                   /* A^(k+2) */
    while (k < 8 && p <= limit) { k++; p *= A; }
    return k;
}

static uint32_t choose_seg(uint64_t n)
{
    uint32_t seg = BQC_SEG_MAX;
    while (seg > 1024 && n / seg < 65536) seg >>= 1;
    return seg;
}

static uint32_t sample_step(uint64_t n)
{
    uint64_t S = n >> 24;
    return S < 1 ? 1u : (S > 64 ? 64u : (uint32_t)S);
}

static void normalise(const uint32_t *cnt, uint32_t A, uint16_t *f)
{
    const uint32_t M = 1u << BQC_SCALE;
    uint64_t T = 0;
    for (uint32_t s = 0; s < A; s++) T += cnt[s];
    uint32_t sum = 0;
    for (uint32_t s = 0; s < A; s++) {
        uint32_t v = T ? (uint32_t)(((uint64_t)cnt[s] * M) / T) : 0;
        if (v == 0) v = 1;
        f[s] = (uint16_t)v; sum += v;
    }
    while (sum > M) {
        uint32_t best = 0;
        for (uint32_t s = 1; s < A; s++) if (f[s] > f[best]) best = s;
        uint32_t d = sum - M;
        if (d > (uint32_t)f[best] - 1u) d = (uint32_t)f[best] - 1u;
        f[best] = (uint16_t)(f[best] - d); sum -= d;
    }
    if (sum < M) {
        uint32_t best = 0;
        for (uint32_t s = 1; s < A; s++) if (f[s] > f[best]) best = s;
        f[best] = (uint16_t)(f[best] + (M - sum));
    }
}

/* (12 - log2 f) * 256 for a frequency f in 1 .. 4096, in integers (the mantissa squared eight times) */
static uint32_t bit_cost(uint32_t f)
{
    uint32_t e = 0;
    while ((2u << e) <= f) e++;
    uint64_t m = (uint64_t)f << (31 - e);                /* [2^31, 2^32) */
    uint32_t frac = 0;
    for (int i = 0; i < 8; i++) {
        m = (m * m) >> 31;
        frac <<= 1;
        if (m >> 32) { frac |= 1; m >>= 1; }
    }
    return BQC_SCALE * 256 - (e * 256 + frac);
}
/* The order the container is made with: the counts were taken at order kmax; for every order k <= kmax (counts summed over the
 * symbols that leave the context) the container's size is estimated -- sum count * cost of the row's normalised frequency,
 * times the sample step, + 16 bits per table entry of a context that occurs + 1 bit per context -- and the smallest wins
 * (the higher order among equals).  A table that costs more than it saves (short streams, data without structure at that
 * depth: the DNA of a collection of little coverage) is not written.  cnt is left holding the counts of the chosen order. */
static uint32_t choose_order(uint32_t *cnt, uint32_t A, uint32_t kmax, uint32_t S)
{
    uint32_t best = kmax;
    uint64_t bestBits = ~0ull;
    uint64_t nctx = 1;
    for (uint32_t j = 0; j < kmax; j++) nctx *= A;
    uint32_t *lvl = (uint32_t *)malloc(nctx * A * 4), *keep = (uint32_t *)malloc(nctx * A * 4);
    if (!lvl || !keep) { free(lvl); free(keep); return kmax; }
    memcpy(lvl, cnt, nctx * A * 4);
    uint16_t f[256];
    for (uint32_t k = kmax;; k--) {
        uint64_t bits = 0, used = 0;
        for (uint64_t c = 0; c < nctx; c++) {
            uint64_t T = 0;
            for (uint32_t s = 0; s < A; s++) T += lvl[c * A + s];
            if (!T) continue;
            used++;
            normalise(lvl + c * A, A, f);
            for (uint32_t s = 0; s < A; s++) bits += (uint64_t)lvl[c * A + s] * bit_cost(f[s]);
        }
        bits = bits / 256 * S + used * A * 16 + nctx;
        if (bits < bestBits) { bestBits = bits; best = k; memcpy(keep, lvl, nctx * A * 4); }
        if (k == 0) break;
        const uint64_t low = nctx / A;                   /* contexts of order k - 1: the k - 1 most recent symbols */
        for (uint64_t c = low; c < nctx; c++)
            for (uint32_t s = 0; s < A; s++) {
                const uint64_t v = (uint64_t)lvl[(c % low) * A + s] + lvl[c * A + s];
                lvl[(c % low) * A + s] = v > 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)v;
            }
        nctx = low;
    }
    uint64_t nb = 1;
    for (uint32_t j = 0; j < best; j++) nb *= A;
    memcpy(cnt, keep, nb * A * 4);
    free(lvl); free(keep);
    return best;
}

/* returns the container's length, -1 when `cap` is too small, -2 on allocation failure */
static int64_t rans_encode(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap)
{
    uint32_t present[256] = {0}, map[256] = {0};
    for (uint64_t i = 0; i < n; i++) present[in[i]] = 1;
    uint8_t alphabet[256] = {0};
    uint32_t A = 0;
    for (uint32_t b = 0; b < 256; b++) if (present[b]) { map[b] = A; alphabet[A++] = (uint8_t)b; }
    if (A == 0) A = 1;                                  /* empty input: one dummy symbol */
    const uint32_t S = sample_step(n);
    uint32_t k = choose_k(A, n / S);                    /* the highest order considered: the counts are taken there */
    uint64_t nctx = 1;
    for (uint32_t j = 0; j < k; j++) nctx *= A;
    uint64_t top = nctx;                          /* A^k: weight of the symbol that leaves the context */
    const uint32_t BQC_SEG = choose_seg(n);
    const uint32_t nseg = (uint32_t)((n + BQC_SEG - 1) / BQC_SEG);
    uint32_t *cnt = (uint32_t *)calloc(nctx * A, 4);
    uint16_t *freq = (uint16_t *)calloc(nctx * A, 2), *cum = (uint16_t *)calloc(nctx * A, 2);
    uint8_t *tmp = (uint8_t *)malloc(2 * BQC_SEG_MAX + 16);
    if (!cnt || !freq || !cum || !tmp) { free(cnt); free(freq); free(cum); free(tmp); return -2; }
    for (uint32_t g = 0; g < nseg; g += S) {
        const uint64_t b = (uint64_t)g * BQC_SEG, e = (b + BQC_SEG < n) ? b + BQC_SEG : n;
        uint64_t ctx = 0;
        for (uint64_t i = b; i < e; i++) {
            const uint32_t s = map[in[i]];
            cnt[ctx * A + s]++;
            const uint32_t outgoing = (k && i >= b + k) ? map[in[i - k]] : 0;
            ctx = k ? ctx * A + s - (uint64_t)outgoing * top : 0;
        }
    }
    k = choose_order(cnt, A, k, S);
    nctx = 1;
    for (uint32_t j = 0; j < k; j++) nctx *= A;
    top = nctx;
    uint64_t nused = 0;
    uint32_t cnt0[256] = {0};
    uint16_t dflt[256] = {0};
    uint8_t *usedc = (uint8_t *)calloc(nctx, 1);
    if (!usedc) { free(cnt); free(freq); free(cum); free(tmp); return -2; }
    for (uint64_t c = 0; c < nctx; c++)
        for (uint32_t s = 0; s < A; s++) { uint64_t v = (uint64_t)cnt0[s] + cnt[c * A + s]; cnt0[s] = v > 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)v; }
    normalise(cnt0, A, dflt);
    for (uint64_t c = 0; c < nctx; c++) {
        uint64_t T = 0;
        for (uint32_t s = 0; s < A; s++) T += cnt[c * A + s];
        if (T) { nused++; usedc[c] = 1; normalise(cnt + c * A, A, freq + c * A); }
        else memcpy(freq + c * A, dflt, 2 * A);
        uint32_t acc = 0;
        for (uint32_t s = 0; s < A; s++) { cum[c * A + s] = (uint16_t)acc; acc += freq[c * A + s]; }
    }
    const uint64_t hdr = BQC_HDR + 256 + 2 * A + (nctx + 7) / 8 + nused * A * 2 + (uint64_t)nseg * 4;
    int64_t ret = -1;
    if (hdr <= cap) {
        uint8_t *p = out;
        memcpy(p, "BFQRANS2", 8); p += 8;
        put64(p, n); p += 8;
        put32(p, BQC_SEG); put32(p + 4, nseg); put32(p + 8, A); put32(p + 12, k); put32(p + 16, BQC_SCALE); p += 20;
        put64(p, orc_codec_checksum(in, n)); p += 8;
        memcpy(p, alphabet, 256); p += 256;
        for (uint32_t s = 0; s < A; s++) { p[0] = (uint8_t)dflt[s]; p[1] = (uint8_t)(dflt[s] >> 8); p += 2; }
        memset(p, 0, (nctx + 7) / 8);
        uint8_t *rows = p + (nctx + 7) / 8;
        for (uint64_t c = 0; c < nctx; c++) {
            if (!usedc[c]) continue;
            p[c >> 3] |= (uint8_t)(1u << (c & 7));
            for (uint32_t s = 0; s < A; s++) { rows[0] = (uint8_t)freq[c * A + s]; rows[1] = (uint8_t)(freq[c * A + s] >> 8); rows += 2; }
        }
        uint8_t *segtab = rows, *pay = rows + (uint64_t)nseg * 4;
        uint64_t used = hdr;
        ret = 0;
        for (uint32_t g = 0; g < nseg && ret == 0; g++) {
            const uint64_t b = (uint64_t)g * BQC_SEG, e = (b + BQC_SEG < n) ? b + BQC_SEG : n;
            /* context in front of the last symbol: the k symbols before it, inside the segment */
            uint64_t ctx = 0;
            const uint64_t len = e - b;
            for (uint64_t i = (len - 1 > k ? e - 1 - k : b); i + 1 < e; i++) ctx = k ? (ctx * A + map[in[i]]) % top : 0;
            uint8_t *q = tmp + 2 * BQC_SEG + 16;
            uint32_t x = BQC_L;
            for (uint64_t i = e; i-- > b;) {
                const uint32_t s = map[in[i]];
                const uint32_t f = freq[ctx * A + s], c0 = cum[ctx * A + s];
                const uint32_t xmax = ((BQC_L >> BQC_SCALE) << 8) * f;
                while (x >= xmax) { *--q = (uint8_t)x; x >>= 8; }
                x = ((x / f) << BQC_SCALE) + (x % f) + c0;
                if (k && i > b) {                                   /* context in front of symbol i-1 */
                    const uint32_t prev = map[in[i - 1]];
                    const uint32_t incoming = (i - 1 >= b + k) ? map[in[i - 1 - k]] : 0;
                    ctx = (ctx + (uint64_t)incoming * top - prev) / A;
                }
            }
            q -= 4; put32(q, x);
            const uint64_t bytes = (uint64_t)(tmp + 2 * BQC_SEG + 16 - q);
            if (used + bytes > cap) { ret = -1; break; }
            memcpy(pay, q, bytes); pay += bytes; used += bytes;
            put32(segtab + 4ull * g, (uint32_t)bytes);
        }
        if (ret == 0) ret = (int64_t)used;
    }
    free(cnt); free(freq); free(cum); free(tmp); free(usedc);
    return ret;
}

/* bytes of the first container of a buffer that may hold several back to back (-1: not one) */
static int64_t dnac_member_len(const uint8_t *in, uint64_t len);
int64_t orc_codec_member_len(const uint8_t *in, uint64_t len)
{
    if (len >= 8 && !memcmp(in, "BFQDNAC1", 8)) return dnac_member_len(in, len);
    if (len >= 32 && !memcmp(in, "BFQLINE1", 8)) { int64_t r = orc_codec_member_len(in + 32, len - 32); return r < 0 ? -1 : r + 32; }
    if (len < BQC_HDR + 256 || memcmp(in, "BFQRANS2", 8)) return -1;
    const uint32_t nseg = get32(in + 20), A = get32(in + 24), k = get32(in + 28);
    if (A == 0 || A > 256 || k > 8) return -1;
    uint64_t nctx = 1;
    for (uint32_t j = 0; j < k; j++) { nctx *= A; if (nctx > (1u << 22)) return -1; }
    uint64_t pos = BQC_HDR + 256 + 2ull * A;
    if (pos + (nctx + 7) / 8 > len) return -1;
    uint64_t nused = 0;
    for (uint64_t c = 0; c < nctx; c++) nused += (in[pos + (c >> 3)] >> (c & 7)) & 1;
    pos += (nctx + 7) / 8 + nused * A * 2;
    if (pos + 4ull * nseg > len) return -1;
    uint64_t total = pos + 4ull * nseg;
    for (uint32_t g = 0; g < nseg; g++) total += get32(in + pos + 4ull * g);
    return total <= len ? (int64_t)total : -1;
}

/* raw length of a container (-1: not one) */
int64_t orc_codec_raw_len(const uint8_t *in, uint64_t len)
{
    if (len >= 72 && !memcmp(in, "BFQDNAC1", 8)) return (int64_t)get64(in + 8);
    if (len >= 32 && !memcmp(in, "BFQLINE1", 8)) return (int64_t)get64(in + 8);
    if (len < BQC_HDR + 256 || memcmp(in, "BFQRANS2", 8)) return -1;
    return (int64_t)get64(in + 8);
}

/* returns raw_len, -1 on a malformed container / short `cap` */
static int64_t rans_decode(const uint8_t *in, uint64_t len, uint8_t *out, uint64_t cap)
{
    if (len < BQC_HDR + 256 || memcmp(in, "BFQRANS2", 8)) return -1;
    const uint64_t n = get64(in + 8);
    const uint32_t seg = get32(in + 16), nseg = get32(in + 20), A = get32(in + 24), k = get32(in + 28), scale = get32(in + 32);
    if (n > cap || seg != choose_seg(n) || scale != BQC_SCALE || A == 0 || A > 256 || k > 8 || nseg != (n + seg - 1) / seg) return -1;
    const uint8_t *alphabet = in + BQC_HDR;
    uint64_t nctx = 1;
    for (uint32_t j = 0; j < k; j++) { nctx *= A; if (nctx > (1u << 22)) return -1; }
    const uint64_t top = nctx;
    if (BQC_HDR + 256 + 2ull * A + (nctx + 7) / 8 > len) return -1;
    const uint8_t *dfl = in + BQC_HDR + 256;
    const uint8_t *used = dfl + 2ull * A;
    const uint8_t *rows = used + (nctx + 7) / 8;
    uint16_t *freq = (uint16_t *)calloc(nctx * A, 2), *cum = (uint16_t *)calloc(nctx * A, 2);
    if (!freq || !cum) { free(freq); free(cum); return -1; }
    int64_t ret = (int64_t)n;
    for (uint64_t c = 0; c < nctx && ret >= 0; c++) {
        const uint8_t *row = dfl;
        if ((used[c >> 3] >> (c & 7)) & 1) {
            if ((uint64_t)(rows - in) + 2ull * A > len) { ret = -1; break; }
            row = rows; rows += 2ull * A;
        }
        uint32_t acc = 0;
        for (uint32_t s = 0; s < A; s++) { freq[c * A + s] = (uint16_t)(row[2 * s] | (row[2 * s + 1] << 8)); cum[c * A + s] = (uint16_t)acc; acc += freq[c * A + s]; }
        if (acc != (1u << scale)) ret = -1;
    }
    const uint8_t *segtab = rows;
    if (ret >= 0 && (uint64_t)(segtab - in) + 4ull * nseg > len) ret = -1;
    const uint8_t *pay = segtab + 4ull * nseg;
    for (uint32_t g = 0; g < nseg && ret >= 0; g++) {
        const uint64_t bytes = get32(segtab + 4ull * g);
        if (bytes < 4 || (uint64_t)(pay - in) + bytes > len) { ret = -1; break; }
        const uint8_t *q = pay, *qe = pay + bytes;
        uint32_t x = get32(q); q += 4;
        const uint64_t b = (uint64_t)g * seg, e = (b + seg < n) ? b + seg : n;
        uint64_t ctx = 0;
        uint32_t hist[8] = {0};
        for (uint64_t i = b; i < e; i++) {
            const uint32_t slot = x & ((1u << scale) - 1);
            const uint16_t *fr = freq + ctx * A, *cu = cum + ctx * A;
            uint32_t s = 0;
            while (s < A && !(fr[s] && slot < (uint32_t)cu[s] + fr[s])) s++;       /* the rows partition [0, 2^scale) */
            if (s == A) { ret = -1; break; }
            x = fr[s] * (x >> scale) + slot - cu[s];
            while (x < BQC_L) { if (q >= qe) { ret = -1; break; } x = (x << 8) | *q++; }
            if (ret < 0) break;
            out[i] = alphabet[s];
            if (k) {
                const uint32_t outgoing = (i >= b + k) ? hist[(i - b) % k] : 0;
                hist[(i - b) % k] = s;
                ctx = ctx * A + s - (uint64_t)outgoing * top;
            }
        }
        pay += bytes;
    }
    free(freq); free(cum);
    if (ret >= 0 && orc_codec_checksum(out, n) != get64(in + 36)) ret = -1;
    return ret;
}


/* ---- read-order DNA: block-adaptive hashed order-16 model ------------------------------------------------------ */
#define DNAC_S 1024u
#define DNAC_W 64u
#define DNAC_KPLUS 1u
#define DNAC_TSKIP 8u
#define DNAC_HDR 72u
static uint32_t dnac_H(uint64_t nbases)
{
    uint32_t H = 12;
    while (H < 32 && (1ull << H) < nbases) H++;
    return H;
}
static uint32_t dnac_K(uint64_t nbases)
{
    uint32_t l4 = 0;
    while (l4 < 32 && (1ull << (2 * l4)) < nbases) l4++;
    const uint32_t K = l4 + DNAC_KPLUS;
    return K < 10 ? 10 : K > 20 ? 20 : K;
}
typedef struct { uint32_t K, H, W, cap, tskip; uint64_t M; } dnac_par;
static dnac_par dnac_make(uint32_t K, uint32_t H, uint32_t W, uint32_t ts) { dnac_par P = {K, H, W, 4080u / W, ts ? ts : 0xFFFFu, (1ull << (3 * K)) - 1}; return P; }
static uint32_t dnac_count(uint64_t row, uint32_t s, const dnac_par *P) { const uint32_t v = (uint32_t)(row >> (12 * s)) & 0xFFFu; return v > P->cap ? P->cap : v; }
static uint64_t dnac_block_segs(uint32_t b, uint64_t nseg)
{
    uint64_t cap = nseg / 64;
    cap = cap < 256 ? 256 : cap > 65536 ? 65536 : cap;
    const uint64_t v = 16ull << (b < 12 ? b : 12);
    return v > cap ? cap : v;
}
static uint64_t dnac_row(uint64_t ctx, uint32_t kk, uint32_t H) { return mix64(ctx * 32 + kk) >> (64 - H); }
static void dnac_freqs(uint64_t row, const dnac_par *P, uint32_t *f)
{
    static const uint32_t prior[5] = {3, 3, 3, 3, 1};
    uint32_t v[5], T = 0, sum = 0, best = 0;
    for (int s = 0; s < 5; s++) { v[s] = dnac_count(row, (uint32_t)s, P) * P->W + prior[s]; T += v[s]; }
    for (int s = 0; s < 5; s++) { f[s] = v[s] * ((1u << BQC_SCALE) - 5u) / T + 1; sum += f[s]; if (f[s] > f[best]) best = (uint32_t)s; }
    f[best] += (1u << BQC_SCALE) - sum;
}
static uint32_t dnac_push(uint64_t row, const dnac_par *P, uint32_t j, uint32_t c)
{
    if (j < P->K) return c;
    uint32_t cnt[5], m = 0, tot = 0;
    for (uint32_t s = 0; s < 5; s++) { cnt[s] = dnac_count(row, s, P); tot += cnt[s]; }
    for (uint32_t s = 1; s < 4; s++) if (cnt[s] > cnt[m]) m = s;
    return (cnt[m] >= 3 && cnt[c] == 0 && tot - cnt[m] <= cnt[m] / 8u) ? m : c;
}
static int dnac_sym(uint8_t b) { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : b == 'N' ? 4 : -1; }
/* 1 when the stream is lines of ACGTN (at least one line, the last one terminated) */
static int dnac_applies(const uint8_t *in, uint64_t n, uint64_t *nreads)
{
    if (n < 65536 || in[n - 1] != '\n') return 0;                 /* (a short stream: the table would still be empty at its end) */
    uint64_t nl = 0, start = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (in[i] == '\n') { if (i - start > 0xFFFFu) return 0; nl++; start = i + 1; }
        else if (dnac_sym(in[i]) < 0) return 0;
    }
    *nreads = nl;
    return nl * 16 <= n;                                          /* lines of 15 bases or more on average, none beyond 65535 */
}
/* the table takes block [r0, r1) in: actual symbols sym[], pushed symbols psh[] (both indexed by base), read offsets boff[] */
static void dnac_update(uint64_t *T, const dnac_par *P, const uint8_t *sym, const uint8_t *psh, const uint64_t *boff, uint64_t r0, uint64_t r1)
{
    const uint32_t K = P->K;
    for (uint64_t r = r0; r < r1; r++) {
        uint64_t ctx = 0;
        const uint64_t b = boff[r], len = boff[r + 1] - b;
        for (uint64_t j = 0; j < len; j++) {
            const uint32_t kk = j < K ? (uint32_t)j : K;
            const int skip = (psh[b + j] & 8) != 0;                /* bit 3 of a pushed symbol: the frozen row knew the base */
            if (!skip) T[dnac_row(ctx, kk, P->H)] += 1ull << (12 * sym[b + j]);
            ctx = ((ctx << 3) | (psh[b + j] & 7)) & P->M;
            if (j >= K && !skip) {
                uint64_t c2 = 0;
                int ok = (psh[b + j - K] & 7) < 4;
                for (uint32_t t = 0; t < K; t++) { const uint8_t x = psh[b + j - t] & 7; if (x > 3) ok = 0; c2 = (c2 << 3) | (uint64_t)(3 - (x & 3)); }
                if (ok) T[dnac_row(c2, K, P->H)] += 1ull << (12 * (3 - (psh[b + j - K] & 7)));
            }
        }
    }
}
/* first read of segment g: the first read whose first base has an index >= g S (nreads when there is none) */
static uint64_t dnac_seg_first(const uint64_t *boff, uint64_t nreads, uint64_t g)
{
    uint64_t lo = 0, hi = nreads;
    while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (boff[mid] >= g * DNAC_S) hi = mid; else lo = mid + 1; }
    return lo;
}
static int64_t dnac_encode(const uint8_t *in, uint64_t n, uint64_t nreads, uint8_t *out, uint64_t cap)
{
    const uint64_t nbases = n - nreads;
    uint64_t *boff = (uint64_t *)malloc((nreads + 1) * 8);
    uint8_t *sym = (uint8_t *)malloc(nbases + 1), *psh = (uint8_t *)malloc(nbases + 1), *lens = (uint8_t *)malloc(nreads * 4 + 4);
    const dnac_par P = dnac_make(dnac_K(nbases), dnac_H(nbases), DNAC_W, DNAC_TSKIP);
    uint64_t *T = (uint64_t *)calloc(1ull << P.H, 8);
    uint32_t *fc = NULL;
    uint8_t *tmp = NULL;
    int64_t ret = -2;
    if (!boff || !sym || !psh || !lens || !T) goto done;
    {
        uint64_t r = 0, o = 0, start = 0;
        for (uint64_t i = 0; i < n; i++) {
            if (in[i] == '\n') { boff[r] = o - (i - start); put32(lens + 4 * r, (uint32_t)(i - start)); r++; start = i + 1; }
            else sym[o++] = (uint8_t)dnac_sym(in[i]);
        }
        boff[nreads] = nbases;
    }
    const uint64_t nseg = (nbases + DNAC_S - 1) / DNAC_S;
    if (nseg > 0xFFFFFFFFull) { ret = -1; goto done; }
    ret = -1;
    if (cap < DNAC_HDR) goto done;
    memcpy(out, "BFQDNAC1", 8); put64(out + 8, n); put64(out + 16, nreads); put64(out + 24, nbases);
    put32(out + 32, P.K); put32(out + 36, P.H); put32(out + 40, DNAC_S); put32(out + 44, (uint32_t)nseg); put32(out + 48, BQC_SCALE); put32(out + 52, P.W | (DNAC_TSKIP << 8));
    put64(out + 56, orc_codec_checksum(in, n));
    const int64_t ll = rans_encode(lens, nreads * 4, out + DNAC_HDR, cap - DNAC_HDR);
    if (ll < 0) { ret = ll; goto done; }
    put64(out + 64, (uint64_t)ll);
    uint64_t used = DNAC_HDR + (uint64_t)ll + 4 * nseg;
    if (used > cap) goto done;
    uint8_t *segtab = out + DNAC_HDR + ll, *pay = segtab + 4 * nseg;
    uint64_t maxseg = 0;
    for (uint64_t g = 0; g < nseg; g++) {
        const uint64_t a = dnac_seg_first(boff, nreads, g), b = dnac_seg_first(boff, nreads, g + 1);
        if (boff[b] - boff[a] > maxseg) maxseg = boff[b] - boff[a];
    }
    fc = (uint32_t *)malloc((maxseg + 1) * 4);
    tmp = (uint8_t *)malloc(2 * maxseg + 16);
    if (!fc || !tmp) { ret = -2; goto done; }
    ret = 0;
    for (uint64_t g0 = 0, blk = 0; g0 < nseg && ret == 0; blk++) {
        uint64_t g1 = g0 + dnac_block_segs((uint32_t)(blk > 0xFFFF ? 0xFFFF : blk), nseg);
        if (g1 > nseg) g1 = nseg;
        for (uint64_t g = g0; g < g1 && ret == 0; g++) {
            const uint64_t ra = dnac_seg_first(boff, nreads, g), rb = dnac_seg_first(boff, nreads, g + 1);
            const uint64_t b0 = boff[ra], cnt = boff[rb] - b0;
            if (!cnt) { put32(segtab + 4 * g, 0); continue; }
            for (uint64_t r = ra; r < rb; r++) {                     /* forward: frequencies and pushed symbols */
                uint64_t ctx = 0;
                const uint64_t len = boff[r + 1] - boff[r];
                for (uint64_t j = 0; j < len; j++) {
                    const uint64_t i = boff[r] + j;
                    const uint32_t kk = j < P.K ? (uint32_t)j : P.K;
                    const uint64_t row = T[dnac_row(ctx, kk, P.H)];
                    uint32_t f[5], cum = 0;
                    dnac_freqs(row, &P, f);
                    for (uint32_t s = 0; s < sym[i]; s++) cum += f[s];
                    fc[i - b0] = f[sym[i]] | (cum << 16);
                    psh[i] = (uint8_t)dnac_push(row, &P, (uint32_t)j, sym[i]);
                    ctx = ((ctx << 3) | psh[i]) & P.M;
                    if (dnac_count(row, sym[i], &P) >= P.tskip) psh[i] |= 8;
                }
            }
            uint8_t *q = tmp + 2 * maxseg + 16;
            uint32_t x = BQC_L;
            for (uint64_t i = cnt; i-- > 0;) {
                const uint32_t f = fc[i] & 0xFFFFu, c0 = fc[i] >> 16;
                const uint32_t xmax = ((BQC_L >> BQC_SCALE) << 8) * f;
                while (x >= xmax) { *--q = (uint8_t)x; x >>= 8; }
                x = ((x / f) << BQC_SCALE) + (x % f) + c0;
            }
            q -= 4; put32(q, x);
            const uint64_t bytes = (uint64_t)(tmp + 2 * maxseg + 16 - q);
            if (used + bytes > cap) { ret = -1; break; }
            memcpy(pay, q, bytes); pay += bytes; used += bytes;
            put32(segtab + 4 * g, (uint32_t)bytes);
        }
        if (ret == 0) dnac_update(T, &P, sym, psh, boff, dnac_seg_first(boff, nreads, g0), dnac_seg_first(boff, nreads, g1));
        g0 = g1;
    }
    if (ret == 0) ret = (int64_t)used;
done:
    free(boff); free(sym); free(psh); free(lens); free(T); free(fc); free(tmp);
    return ret;
}
static int64_t rans_decode(const uint8_t *in, uint64_t len, uint8_t *out, uint64_t cap);
static int64_t dnac_member_len(const uint8_t *in, uint64_t len)
{
    if (len < DNAC_HDR || memcmp(in, "BFQDNAC1", 8)) return -1;
    const uint64_t nseg = get32(in + 44), ll = get64(in + 64);
    if (ll > len || DNAC_HDR + ll + 4 * nseg > len) return -1;
    uint64_t total = DNAC_HDR + ll + 4 * nseg;
    for (uint64_t g = 0; g < nseg; g++) total += get32(in + DNAC_HDR + ll + 4 * g);
    return total <= len ? (int64_t)total : -1;
}
static int64_t dnac_decode(const uint8_t *in, uint64_t len, uint8_t *out, uint64_t cap)
{
    if (dnac_member_len(in, len) < 0) return -1;
    const uint64_t n = get64(in + 8), nreads = get64(in + 16), nbases = get64(in + 24), ll = get64(in + 64);
    const uint32_t K = get32(in + 32), H = get32(in + 36), S = get32(in + 40), nseg = get32(in + 44), scale = get32(in + 48);
    const uint32_t W = get32(in + 52) & 0xFF, ts = (get32(in + 52) >> 8) & 0xFF;
    if (n > cap || nreads > n || nbases != n - nreads || K < 8 || K > 20 || H < 12 || H > 32 || S != DNAC_S || scale != BQC_SCALE || W < 1 || W > 64 || (get32(in + 52) >> 16) ||
        nseg != (nbases + DNAC_S - 1) / DNAC_S || nreads == 0) return -1;
    const dnac_par P = dnac_make(K, H, W, ts);
    if (orc_codec_raw_len(in + DNAC_HDR, ll) != (int64_t)(nreads * 4)) return -1;
    uint8_t *lens = (uint8_t *)malloc(nreads * 4 + 4);
    uint64_t *boff = (uint64_t *)malloc((nreads + 1) * 8);
    uint8_t *sym = (uint8_t *)malloc(nbases + 1), *psh = (uint8_t *)malloc(nbases + 1);
    uint64_t *T = (uint64_t *)calloc(1ull << H, 8);
    int64_t ret = -1;
    if (!lens || !boff || !sym || !psh || !T) goto done;
    if (rans_decode(in + DNAC_HDR, ll, lens, nreads * 4) != (int64_t)(nreads * 4)) goto done;
    {
        uint64_t acc = 0;
        for (uint64_t r = 0; r < nreads; r++) { boff[r] = acc; acc += get32(lens + 4 * r); }
        boff[nreads] = acc;
        if (acc != nbases) goto done;
    }
    const uint8_t *segtab = in + DNAC_HDR + ll, *pay = segtab + 4ull * nseg;
    ret = (int64_t)n;
    for (uint64_t g0 = 0, blk = 0; g0 < nseg && ret >= 0; blk++) {
        uint64_t g1 = g0 + dnac_block_segs((uint32_t)(blk > 0xFFFF ? 0xFFFF : blk), nseg);
        if (g1 > nseg) g1 = nseg;
        for (uint64_t g = g0; g < g1 && ret >= 0; g++) {
            const uint64_t ra = dnac_seg_first(boff, nreads, g), rb = dnac_seg_first(boff, nreads, g + 1);
            const uint64_t bytes = get32(segtab + 4 * g);
            if (boff[rb] == boff[ra]) { if (bytes) ret = -1; continue; }
            if (bytes < 4) { ret = -1; break; }
            const uint8_t *q = pay, *qe = pay + bytes;
            uint32_t x = get32(q); q += 4;
            for (uint64_t r = ra; r < rb && ret >= 0; r++) {
                uint64_t ctx = 0;
                const uint64_t rl = boff[r + 1] - boff[r];
                for (uint64_t j = 0; j < rl; j++) {
                    const uint64_t i = boff[r] + j;
                    const uint32_t kk = j < P.K ? (uint32_t)j : P.K;
                    const uint64_t row = T[dnac_row(ctx, kk, P.H)];
                    uint32_t f[5], cum = 0, s = 0;
                    dnac_freqs(row, &P, f);
                    const uint32_t slot = x & ((1u << BQC_SCALE) - 1);
                    while (s < 4 && slot >= cum + f[s]) { cum += f[s]; s++; }
                    x = f[s] * (x >> BQC_SCALE) + slot - cum;
                    while (x < BQC_L) { if (q >= qe) { ret = -1; break; } x = (x << 8) | *q++; }
                    if (ret < 0) break;
                    sym[i] = (uint8_t)s;
                    psh[i] = (uint8_t)dnac_push(row, &P, (uint32_t)j, s);
                    ctx = ((ctx << 3) | psh[i]) & P.M;
                    if (dnac_count(row, s, &P) >= P.tskip) psh[i] |= 8;
                }
            }
            pay += bytes;
        }
        if (ret >= 0) dnac_update(T, &P, sym, psh, boff, dnac_seg_first(boff, nreads, g0), dnac_seg_first(boff, nreads, g1));
        g0 = g1;
    }
    if (ret >= 0) {
        uint64_t o = 0;
        for (uint64_t r = 0; r < nreads; r++) {
            for (uint64_t i = boff[r]; i < boff[r + 1]; i++) out[o++] = (uint8_t)"ACGTN"[sym[i]];
            out[o++] = '\n';
        }
        if (orc_codec_checksum(out, n) != get64(in + 56)) ret = -1;
    }
done:
    free(lens); free(boff); free(sym); free(psh); free(T);
    return ret;
}

/* ---- line-delta transform -------------------------------------------------------------------------------------- */
#define BQC_LINE_R 256u
/* transformed length, or 0 when the stream is not made of '\n'-terminated lines (or has fewer than two) */
static uint64_t line_xform(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t *nlines)
{
    if (n < 2 || in[n - 1] != '\n') return 0;
    uint64_t o = 0, prev = 0, prevLen = 0, i = 0, ln = 0;
    while (i < n) {
        uint64_t e = i;
        while (in[e] != '\n') e++;
        const uint64_t len = e - i + 1;
        uint64_t p = 0;
        if (ln % BQC_LINE_R) {
            uint64_t lim = len - 1;
            if (lim > prevLen) lim = prevLen;
            if (lim > 254) lim = 254;
            while (p < lim && in[i + p] == in[prev + p]) p++;
        }
        if (out) { out[o] = (uint8_t)(p + (p >= 10)); memcpy(out + o + 1, in + i + p, len - p); }
        o += 1 + len - p;
        prev = i; prevLen = len; i = e + 1; ln++;
    }
    *nlines = ln;
    return (ln >= 2 && n / ln >= 8 && n / ln <= 128) ? o : 0;   /* read names: lines of 8 .. 128 bytes on average */
}

int64_t orc_codec_encode(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap)
{
    uint64_t nl = 0;
    if (dnac_applies(in, n, &nl)) {
        /* little coverage (a sample of a large genome) leaves the context table nothing to learn and its chance hits cost:
         * above 1.6 bits per base the plain static container is made as well, and the smaller of the two is the output */
        const int64_t r = dnac_encode(in, n, nl, out, cap);
        if (r < 0 || 5 * (uint64_t)r <= n - nl) return r;
        uint8_t *t = (uint8_t *)malloc(cap);
        if (!t) return -2;
        const int64_t r2 = rans_encode(in, n, t, cap);
        if (r2 >= 0 && r2 < r) memcpy(out, t, (size_t)r2);
        free(t);
        return (r2 >= 0 && r2 < r) ? r2 : r;
    }
    const uint64_t xl = line_xform(in, n, NULL, &nl);
    if (!xl || xl * 4 > n * 3) return rans_encode(in, n, out, cap);
    if (cap < 32) return -1;
    uint8_t *t = (uint8_t *)malloc(xl);
    if (!t) return -2;
    line_xform(in, n, t, &nl);
    memcpy(out, "BFQLINE1", 8); put64(out + 8, n); put32(out + 16, BQC_LINE_R); put32(out + 20, 0); put64(out + 24, nl);
    const int64_t r = rans_encode(t, xl, out + 32, cap - 32);
    free(t);
    return r < 0 ? r : r + 32;
}

int64_t orc_codec_decode(const uint8_t *in, uint64_t len, uint8_t *out, uint64_t cap)
{
    if (len >= 8 && !memcmp(in, "BFQDNAC1", 8)) return dnac_decode(in, len, out, cap);
    if (len < 32 || memcmp(in, "BFQLINE1", 8)) return rans_decode(in, len, out, cap);
    const uint64_t n = get64(in + 8), nl = get64(in + 24);
    const uint32_t R = get32(in + 16);
    if (n > cap || R == 0) return -1;
    const int64_t xl = orc_codec_raw_len(in + 32, len - 32);
    if (xl < 0) return -1;
    uint8_t *t = (uint8_t *)malloc((size_t)xl + 1);
    if (!t) return -1;
    int64_t ret = (int64_t)n;
    if (rans_decode(in + 32, len - 32, t, (uint64_t)xl) != xl) ret = -1;
    uint64_t i = 0, o = 0, prev = 0, ln = 0;
    while (ret >= 0 && i < (uint64_t)xl) {
        uint64_t p = t[i];
        if (p == 10) { ret = -1; break; }
        if (p > 10) p--;
        uint64_t e = i + 1;
        while (e < (uint64_t)xl && t[e] != '\n') e++;
        if (e >= (uint64_t)xl || o + p + (e - i) > n || (p && (ln % R == 0 || prev + p > o))) { ret = -1; break; }
        memmove(out + o, out + prev, p);
        memcpy(out + o + p, t + i + 1, e - i);
        prev = o; o += p + (e - i); i = e + 1; ln++;
    }
    if (ret >= 0 && (o != n || ln != nl)) ret = -1;
    free(t);
    return ret;
}
