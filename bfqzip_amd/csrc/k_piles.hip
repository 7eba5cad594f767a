// k_piles.hip -- step 1 one pile at a time (first stage): the suffixes are partitioned by their
// first symbol -- the "piles" bfq_ext keeps as files (bwt_<p>.aux, bfq_ext.cpp:190-348) -- and every pile is sorted,
// refined and emitted on its own; the outputs concatenate in the order # A C G N T (dna_bwt_n.hpp:46-61).
//
// Why: the sort records (24 B per suffix, ping-pong) are the bulk of step 1's workspace.  With piles only the largest
// pile's records are resident at a time: 13 n bytes instead of 28.6 n (30 M x 150 bp: 55 GiB instead of 121 GiB), so a
// GPU takes blocks about twice as large, and the drop-in tools, which allocate their workspace once per process, start
// faster.  Same kernels (k_radix_*, k_refine_*, k_huge_*, k_emit_bwt) on pile-local arrays; new here:
//   k_pile_count      : symbol counts per block of text positions                          (1 B/position read)
//   k_build_keys_pile : ordered compaction of one pile's suffixes into sort records        (~1.3 B/position + 12 B/record)
//   k_term_pile       : the pile of the N terminator suffixes "#_i": rows 0..N-1 in read order, nothing to sort
// The LCP at a pile's first row is 0 (its first symbol differs from the row before it), which is what bfq_refine writes.
#include "bfq_internal.h"
#include "bfq_device.h"

#define PB BFQ_RS_BLOCK_ELEMS                          // text positions per block

// first == 7: counts of the symbol at every position; else counts of the SECOND symbol of the suffixes that start with `first`
__global__ __launch_bounds__(256) void k_pile_count(const u8 *__restrict__ T8, u64 n, u32 first, u32 *__restrict__ cntBlk, u64 nb)
{
    __shared__ u32 sh[6];
    for (u64 b = blockIdx.x; b < nb; b += gridDim.x) {
        if (threadIdx.x < 6) sh[threadIdx.x] = 0;
        __syncthreads();
        const u64 lo = b * (u64)PB, hi = (lo + PB < n) ? lo + PB : n;
        u32 c[6] = {0, 0, 0, 0, 0, 0};
        for (u64 p = lo + (u64)threadIdx.x * 16; p < hi; p += 256 * 16) {       // T8 is 16-byte aligned (and padded), PB a multiple of 16
            if (p + 16 <= hi && p + 17 <= n + 64) {                    // 16 positions from one 16-byte load (T8 is padded by 64 bytes)
                const uint4 x = *(const uint4 *)(T8 + p);
                const u32 w[4] = {x.x, x.y, x.z, x.w};
                const u32 after = (p + 16 < n) ? (u32)T8[p + 16] : 0u; // symbol that follows the 16th position
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        u32 v = (w[q] >> (8 * k)) & 7u;
                        if (first != 7u) {
                            if (v != first) continue;
                            v = (k < 3) ? (w[q] >> (8 * (k + 1))) & 7u : (q < 3 ? w[q + 1] & 7u : after & 7u);   // its second symbol
                        }
#pragma unroll
                        for (int s = 0; s < 6; s++) c[s] += (v == (u32)s);
                    }
            } else {
                for (u64 q = p; q < hi && q < p + 16; q++) {
                    u32 v = T8[q] & 7u;
                    if (first != 7u) { if (v != first) continue; v = (q + 1 < n) ? (T8[q + 1] & 7u) : 0u; }   // a base is always followed by something
                    for (int s = 0; s < 6; s++) c[s] += (v == (u32)s);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 6; s++) {
            const u32 t = (u32)__builtin_amdgcn_readlane((int)bfq_wave_incscan32(c[s]), 63);
            if ((threadIdx.x & 63) == 0 && t) atomicAdd(&sh[s], t);
        }
        __syncthreads();
        if (threadIdx.x < 6) cntBlk[(u64)threadIdx.x * nb + b] = sh[threadIdx.x];
        __syncthreads();
    }
}

// records of the suffixes that start with symbol `code`, in text order: block b's go to [blkOff[b], blkOff[b + 1])
// (code2 == 7: any second symbol).  16 positions per thread from one 16-byte load each of T8 and Q8 (both 16-byte aligned,
// PB a multiple of 16), one block scan per 4096 positions: the scan of the whole text costs ~3 ms at 4.5 G positions, which
// matters when a collection is cut into 20-25 two-symbol piles (one scan per pile).
__global__ __launch_bounds__(256) void k_build_keys_pile(const u8 *__restrict__ T8, const u8 *__restrict__ Q8, const u64 *__restrict__ text3,
                                                         u64 n, u32 code, u32 code2, const u64 *__restrict__ blkOff, SortRec out, u64 nb)
{
    __shared__ u32 scan[4];
    constexpr u32 SWEEP = 4096;
    const u32 sw = SWEEP / BFQ_SYMS_PER_WORD, so = SWEEP - sw * BFQ_SYMS_PER_WORD;
    for (u64 hb = blockIdx.x; hb < nb; hb += gridDim.x) {
        const u64 bbase = hb * (u64)PB;
        u64 bend = bbase + PB;
        if (bend > n) bend = n;
        u64 dst = blkOff[hb];
        u64 p0 = bbase + (u64)threadIdx.x * 16;
        u64 w0 = p0 / BFQ_SYMS_PER_WORD;
        u32 o0 = (u32)(p0 - w0 * BFQ_SYMS_PER_WORD);
        for (u64 sweep = bbase; sweep < bend; sweep += SWEEP, p0 += SWEEP) {   // uniform trip count: barriers inside
            u32 c[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
            u32 before = 0, after = 0, qbefore = (u32)'#';
            u32 match = 0;
            if (p0 < bend) {
                const uint4 cx = *(const uint4 *)(T8 + p0);              // T8 / Q8 are padded by 64 bytes
                const uint4 qx = *(const uint4 *)(Q8 + p0);
                c[0] = cx.x; c[1] = cx.y; c[2] = cx.z; c[3] = cx.w;
                q[0] = qx.x; q[1] = qx.y; q[2] = qx.z; q[3] = qx.w;
                if (p0 >= 1) { before = T8[p0 - 1]; qbefore = Q8[p0 - 1]; }
                if (p0 + 16 < n) after = T8[p0 + 16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const u32 ck = (c[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                    const u32 nk = k < 15 ? (c[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 0xFFu : after;
                    if (p0 + k < bend && ck == code && (code2 == 7u || nk == code2)) match |= 1u << k;
                }
            }
            u32 tot;
            const u32 ex = bfq_block_exscan32((u32)__popc(match), scan, &tot);
            if (match) {
                const u64 *t3 = text3 + bfq_t3_at(w0);
                const u64 t0 = t3[0], t1 = t3[1], t2 = t3[2];                        // 16 windows span at most 3 words
                const u64 clo = (u64)c[0] | ((u64)c[1] << 32), chi = (u64)c[2] | ((u64)c[3] << 32);
                const u64 qlo = (u64)q[0] | ((u64)q[1] << 32), qhi = (u64)q[2] | ((u64)q[3] << 32);
                u64 d = dst + ex;
                // one matching position per iteration and lane (a lane holds ~1 match of a two-symbol pile, ~4 of a
                // one-symbol pile): the key packing runs max-matches-per-lane times, not 16 times, per wavefront
                for (u32 mm = match; mm; mm &= mm - 1) {
                    const u32 k = (u32)__builtin_ctz(mm);
                    const u32 ok = o0 + k;                                           // <= 35
                    const bool second = ok >= BFQ_SYMS_PER_WORD;
                    const u32 o3 = (second ? ok - BFQ_SYMS_PER_WORD : ok) * 3u;
                    const u64 a = second ? t1 : t0, bnext = second ? t2 : t1;
                    const u64 hi = (a << o3) & BFQ_M63;
                    const u64 lo = o3 ? (bnext >> (63u - o3)) : 0ull;
                    const u64 sk = bfq_skey_of(bfq_mask_key(hi | lo));
                    const u32 kp = k - 1u;                                           // previous position inside the 16 (k > 0)
                    const u32 pc = k ? (u32)(((kp < 8 ? clo : chi) >> (8 * (kp & 7))) & 0xFFu) : before;
                    const u32 pqr = k ? (u32)(((kp < 8 ? qlo : qhi) >> (8 * (kp & 7))) & 0xFFu) : qbefore;
                    const u32 pq = pc ? pqr : (u32)'#';
                    const u64 pay = bfq_pack_val(p0 + k, pc, pq);
                    out.w0[d] = bfq_rec_w0(sk);
                    out.w12[d] = ((u64)bfq_rec_w2(pay) << 32) | bfq_rec_w1(sk, pay);
                    d++;
                }
            }
            dst += tot;
            w0 += sw; o0 += so;
            if (o0 >= BFQ_SYMS_PER_WORD) { o0 -= BFQ_SYMS_PER_WORD; w0++; }
        }
        __syncthreads();
    }
}

// rows 0..N-1: the suffix "#_i" of read i; its eBWT symbol is the read's last base (the terminator for an empty read)
__global__ __launch_bounds__(256) void k_term_pile(const u8 *__restrict__ T8, const u8 *__restrict__ Q8, const u64 *__restrict__ roff, u64 N,
                                                   u32 termOut, u8 *__restrict__ bwt, u8 *__restrict__ qs, u16 *__restrict__ lcp)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        const u64 b = roff[i], e = roff[i + 1];
        const u64 tp = e + i;                                    // text position of the read's terminator
        u32 code = 0, q = (u32)'#';
        if (e > b) { code = T8[tp - 1]; q = Q8[tp - 1]; }
        bwt[i] = code ? bfq_code_sym(code) : (u8)termOut;
        qs[i] = (u8)q;
        if (lcp) lcp[i] = 0;
    }
}

// start of a pile's refinement: the list lengths of the previous pile are folded into the totals
__global__ void k_refine_reset(DevCounters *cnt)
{
    cnt->bigTotal += cnt->bigCount;
    cnt->bigCount = 0; cnt->hugeCount = 0; cnt->hugeRows = 0;
}

// workspace bound of the pile mode (pile records for at most `cap` rows).  lean (the one-shot tools): the eBWT and its
// qualities live outside the arena, the text arrays are given, and the LCP is a per-pile scratch nobody reads.
size_t bfq_ws_need_piles(u64 n, u64 N, u64 cap, u64 extra, bool lean)
{
    u64 nb = n / 32768 + 2, nbc = ceil_div(cap + 1, bfq_radix_block_elems(cap)) + 8200;   // a smaller pile may use smaller radix blocks
    size_t need = 0;
    if (lean) need += 2 * (cap + 256) + 4096;                    // lcp16 of one pile
    else {
        need += 4 * (n + 256) + 4096;                            // bwt, qual, lcp16
        need += 8 * bfq_t3_alloc(n / 21 + 3) + 2 * (n + 256);                // packed text, T8, Q8
    }
    need += 6 * 4 * (cap + 256);                                 // sort records of one pile, ping-pong
    need += 256 * nbc * 12 + (nbc + 4096) * 64;                  // radix histograms + scan partials
    need += 2 * 6 * 12 * nb + 16 * (N + 64);                     // block counts / offsets per symbol (twice: a split pile), read offsets
    need += extra + (64u << 20);
    return need;
}

// pre != nullptr: the text arrays exist already (one-shot tools: they live in an allocation of their own, so that the
// arena can be sized from the actual pile sizes).  capTarget != 0: piles above it are split by their second symbol even
// when they would fit.  c->lcpScratch: nobody wants the LCP -- every pile writes its entries to the same scratch.
void bfq_step1_piles(bfq_ctx *c, const u8 *d_bases, const u8 *d_quals, const u64 *d_roff, u64 N, u64 total, int termOut, bfq_stats *st,
                     const PileText *pre, u64 capTarget)
{
    const u64 n = total + N;
    if (n >= (1ull << BFQ_POS_BITS)) throw BfqError{BFQ_E_ARG, "collection too large (2^37 rows)"};
    c->n = n; c->N = N;
    c->d_bwt = c->extBwt ? c->extBwt : c->alloc<u8>(n + 64);
    c->d_qual = c->extQual ? c->extQual : c->alloc<u8>(n + 64);
    const bool scratch = c->lcpScratch;
    c->d_lcp = scratch ? nullptr : c->alloc<u16>(n + 64);
    c->d_gcnt = nullptr; c->gcntTerm = -1;                       // symbol counts per group are not produced pile by pile
    if (!n) return;
    const size_t m0 = c->mark();
    const u64 nwords = n / BFQ_SYMS_PER_WORD + 3;
    u64 *text3;
    u8 *T8, *Q8;
    if (pre) { text3 = pre->text3; T8 = pre->T8; Q8 = pre->Q8; }
    else {
        text3 = c->alloc<u64>(bfq_t3_alloc(nwords));
        T8 = c->alloc<u8>(n + 64); Q8 = c->alloc<u8>(n + 64);
        bfq_build_text(c, d_bases, d_quals, d_roff, N, n, T8, Q8, text3, nwords);
    }
    const u64 nb = ceil_div(n, PB);
    u32 *cntBlk = c->alloc<u32>(6 * nb);
    u64 *blkOff = c->alloc<u64>(6 * nb + 8);
    u64 *d_tot = c->alloc<u64>(8);
    KLAUNCH(c, K_KEYS, (double)n, k_pile_count, bfq_grid(nb, 1), 256, (const u8 *)T8, n, 7u, cntBlk, nb);
    for (int s = 0; s < 6; s++) bfq_exscan_u32(c, cntBlk + (u64)s * nb, blkOff + (u64)s * nb, nb, d_tot + s);
    u64 tot[6];
    HIP_CHECK(hipMemcpyAsync(tot, d_tot, 48, hipMemcpyDeviceToHost, c->stream));
    c->fetchCounters();                                          // also waits for tot; errSymbol / errTooLong of the text build
    if (c->h_cnt.errSymbol || c->h_cnt.errTooLong) { c->release(m0); return; }   // reported by the caller's check
    if (tot[0] != N) throw BfqError{BFQ_E_ARG, "pile counts do not match the collection"};
    if (N) KLAUNCH(c, K_EMIT, 12.0 * (double)N, k_term_pile, bfq_grid(N, 256), 256, (const u8 *)T8, (const u8 *)Q8, d_roff, N,
                   (u32)(termOut & 0xFF), c->d_bwt, c->d_qual, c->d_lcp);
    if (c->onRows) c->onRows(0, N);

    const size_t avail = c->wsCap - c->wsTop;
    auto fits = [&](u64 m) { return (size_t)(24 * (m + 256) + 12 * 256 * (ceil_div(m + 1, bfq_radix_block_elems(m)) + 8) + (m / 32768 + 4096) * 64 + (scratch ? 2 * (m + 256) : 0) + (48u << 20)) <= avail; };
    // one pile (first symbol s, second symbol s2 or 7 = any) of m suffixes -> rows [start, start + m)
    auto run_pile = [&](u32 s, u32 s2, u64 m, u64 start, const u64 *off) {
        const size_t mp = c->mark();
        SortRec A, B;
        A.w0 = c->alloc<u32>(m + 16); A.w12 = c->alloc<u64>(m + 16);
        const size_t mB = c->mark();
        B.w0 = c->alloc<u32>(m + 16); B.w12 = c->alloc<u64>(m + 16);
        KLAUNCH(c, K_KEYS, 1.3 * (double)n + 12.0 * (double)m, k_build_keys_pile, bfq_grid(nb, 1), 256, (const u8 *)T8, (const u8 *)Q8,
                (const u64 *)text3, n, s, s2, off, B, nb);
        bfq_radix_sort(c, B, A, m);                               // five passes: B -> A
        c->release(mB);
        hipLaunchKernelGGL(k_refine_reset, dim3(1), dim3(1), 0, c->stream, c->d_cnt);
        u16 *lcpOut = scratch ? c->alloc<u16>(m + 64) : c->d_lcp + start;
        bfq_refine(c, A, text3, m, lcpOut, st);
        bfq_emit_bwt(c, A, m, termOut, c->d_bwt + start, c->d_qual + start, nullptr);
        c->release(mp);
    };
    u64 start = N;
    for (u32 s = 1; s <= 5; s++) {
        const u64 m = tot[s];
        if (!m) continue;
        if (fits(m) && !c->env.pilesSplit && !(capTarget && m > capTarget)) { run_pile(s, 7u, m, start, blkOff + (u64)s * nb); if (c->onRows) c->onRows(start, m); start += m; continue; }
        // a pile beyond the workspace (skewed base composition, low-complexity reads): once more by its second symbol
        const size_t ms = c->mark();
        u32 *cnt2 = c->alloc<u32>(6 * nb);
        u64 *off2 = c->alloc<u64>(6 * nb + 8);
        u64 *d_tot2 = c->alloc<u64>(8);
        KLAUNCH(c, K_KEYS, (double)n, k_pile_count, bfq_grid(nb, 1), 256, (const u8 *)T8, n, s, cnt2, nb);
        for (int q = 0; q < 6; q++) bfq_exscan_u32(c, cnt2 + (u64)q * nb, off2 + (u64)q * nb, nb, d_tot2 + q);
        u64 tot2[6];
        HIP_CHECK(hipMemcpyAsync(tot2, d_tot2, 48, hipMemcpyDeviceToHost, c->stream));
        c->sync();
        bool firstSub = true;
        for (u32 s2 = 0; s2 <= 5; s2++) {
            const u64 m2 = tot2[s2];
            if (!m2) continue;
            if (!fits(m2)) {
                char b[220];
                snprintf(b, sizeof b, "pile '%c%c' holds %llu of %llu suffixes: larger than the pile workspace (use the one-piece mode or a larger GPU share)",
                         "#ACGNT"[s], "#ACGNT"[s2], (unsigned long long)m2, (unsigned long long)n);
                throw BfqError{BFQ_E_NOMEM, b};
            }
            run_pile(s, s2, m2, start, off2 + (u64)s2 * nb);
            if (!firstSub && !scratch) {                          // the row before shares exactly the first symbol
                const u16 one = 1;
                HIP_CHECK(hipMemcpyAsync(c->d_lcp + start, &one, 2, hipMemcpyHostToDevice, c->stream));
                c->sync();
            }
            firstSub = false;
            if (c->onRows) c->onRows(start, m2);
            start += m2;
        }
        c->release(ms);
    }
    c->release(m0);
}

// ---- one pile on its own (global mode, k_global.hip): the pile (first symbol s, second symbol s2 or 7 = any) of the text
// T8 / Q8 / text3 is sorted and refined into pile-local arrays; the sorted records' (w1, w2) words stay for the caller
// (every row knows the text position of its suffix).  Everything lives in the arena above the caller's mark.
u64 bfq_run_one_pile(bfq_ctx *c, const u8 *T8, const u8 *Q8, const u64 *text3, u64 n, u32 s, u32 s2, int termOut, PileRows *out)
{
    const u64 nb = ceil_div(n, PB);
    u32 *cntBlk = c->alloc<u32>(6 * nb);
    u64 *blkOff = c->alloc<u64>(6 * nb + 8);
    u64 *d_tot = c->alloc<u64>(8);
    const u32 want = (s2 == 7u) ? s : s2;
    KLAUNCH(c, K_KEYS, (double)n, k_pile_count, bfq_grid(nb, 1), 256, T8, n, (s2 == 7u) ? 7u : s, cntBlk, nb);
    bfq_exscan_u32(c, cntBlk + (u64)want * nb, blkOff, nb, d_tot);
    u64 m = 0;
    HIP_CHECK(hipMemcpyAsync(&m, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    out->m = m;
    if (!m) return 0;
    out->bwt = c->alloc<u8>(m + 64); out->qs = c->alloc<u8>(m + 64); out->lcp = c->alloc<u16>(m + 64);
    SortRec A, B;
    A.w12 = c->alloc<u64>(m + 16);
    const size_t mKeep = c->mark();
    A.w0 = c->alloc<u32>(m + 16);
    const size_t mB = c->mark();
    B.w0 = c->alloc<u32>(m + 16); B.w12 = c->alloc<u64>(m + 16);
    KLAUNCH(c, K_KEYS, 1.3 * (double)n + 12.0 * (double)m, k_build_keys_pile, bfq_grid(nb, 1), 256, T8, Q8, text3, n, s, s2, (const u64 *)blkOff, B, nb);
    bfq_radix_sort(c, B, A, m);                                   // five passes: B -> A
    c->release(mB);
    hipLaunchKernelGGL(k_refine_reset, dim3(1), dim3(1), 0, c->stream, c->d_cnt);
    bfq_refine(c, A, text3, m, out->lcp, nullptr);
    bfq_emit_bwt(c, A, m, termOut, out->bwt, out->qs, nullptr);
    c->release(mKeep);
    out->w12 = A.w12;
    return m;
}

// suffix counts by (first, second) symbol: counts[6 * s + s2]; row s = 0 holds the N terminator suffixes in counts[0]
void bfq_pile_pair_counts(bfq_ctx *c, const u8 *T8, u64 n, u64 *counts36)
{
    const size_t mk = c->mark();
    const u64 nb = ceil_div(n, PB);
    u32 *cntBlk = c->alloc<u32>(6 * nb);
    u64 *off = c->alloc<u64>(6 * nb + 8);
    u64 *d_tot = c->alloc<u64>(40);
    for (u32 f = 0; f <= 5; f++) {
        KLAUNCH(c, K_KEYS, (double)n, k_pile_count, bfq_grid(nb, 1), 256, T8, n, f == 0 ? 7u : f, cntBlk, nb);
        for (int q = 0; q < 6; q++) bfq_exscan_u32(c, cntBlk + (u64)q * nb, off, nb, d_tot + 6 * f + q);
    }
    HIP_CHECK(hipMemcpyAsync(counts36, d_tot, 36 * 8, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    const u64 N = counts36[0];                                   // f == 0 counted single symbols: [#, A, C, G, N, T]
    for (int q = 1; q < 6; q++) counts36[q] = 0;
    counts36[0] = N;
    c->release(mk);
}
