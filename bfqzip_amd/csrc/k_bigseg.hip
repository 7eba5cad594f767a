// k_bigseg.hip -- suffix order inside very large equal-prefix segments (low-complexity reads:
// poly-A / poly-G tails, short tandem repeats), where one workgroup per segment (k_refine_big)
// would serialise millions of rows.
//
// The rows of all such segments of a batch are laid out as "slots" (slot j <-> a fixed eBWT row,
// segments back to back).  One round looks at the next 16 symbols of every slot's suffix:
//   k_huge_keys    : sort key of the suffix at depth d, record (key, slot)
//   radix sort     : by key (the 5-pass LSD sort of step 1, stable)
//   k_huge_segkeys : record (sub-segment id, slot) in that order; radix sort (4 passes, stable)
//                    -> grouped by sub-segment, ordered by key, ties in the previous order
//   k_huge_apply   : payloads / keys gathered into the new slot order
//   k_huge_bounds  : payload written back to its eBWT row, LCP of every new boundary
//                    (d + common prefix of the two keys); slots still tied with a neighbour
//                    (equal keys without a terminator) are flagged
//   scans + k_huge_compact : the tied slots, with new dense sub-segment ids, form the next round
// until nothing is tied.  Every step is a flat, coalesced pass over the slots, so a segment of
// any size uses the whole device.  Identical suffixes (equal keys holding the terminator) keep
// their position order because every sort is stable.
// Order and LCP conventions as in k_refine.hip.
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_rec.h"
#include <algorithm>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k_huge_expand(const u64 *__restrict__ hstart, const u64 *__restrict__ hoff, u32 nseg, u64 m,
                                                     SortRec rec, u64 *__restrict__ grow, u32 *__restrict__ gseg,
                                                     u64 *__restrict__ pay)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        u32 lo = 0, hi = nseg;                         // last segment with hoff[seg] <= j
        while (hi - lo > 1) {
            u32 mid = (lo + hi) >> 1;
            if (hoff[mid] <= j) lo = mid; else hi = mid;
        }
        u64 r = hstart[lo] + (j - hoff[lo]);
        grow[j] = r; gseg[j] = lo; pay[j] = rec_pay(rec, r);
    }
}

__global__ __launch_bounds__(256) void k_huge_keys(const u64 *__restrict__ pay, const u64 *__restrict__ text3, u64 m, u32 depth,
                                                   u64 *__restrict__ K, SortRec R)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        u64 k = bfq_skey_of(bfq_key_at(text3, bfq_val_pos(pay[j]) + depth));
        K[j] = k;
        R.w0[j] = bfq_rec_w0(k);
        R.w12[j] = ((u64)bfq_rec_w2(j) << 32) | bfq_rec_w1(k, j);
    }
}

__global__ __launch_bounds__(256) void k_huge_segkeys(SortRec Rin, const u32 *__restrict__ gseg, u64 m, SortRec Rout)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < m; t += stride) {
        u64 idx = rec_pay(Rin, t);
        u64 s = gseg[idx];
        Rout.w0[t] = bfq_rec_w0(s);
        Rout.w12[t] = ((u64)bfq_rec_w2(idx) << 32) | bfq_rec_w1(s, idx);
    }
}

__global__ __launch_bounds__(256) void k_huge_apply(SortRec R2, const u64 *__restrict__ pay, const u64 *__restrict__ K, u64 m,
                                                    u64 *__restrict__ npay, u64 *__restrict__ nK)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        u64 idx = rec_pay(R2, j);
        npay[j] = pay[idx]; nK[j] = K[idx];
    }
}

__global__ __launch_bounds__(256) void k_huge_bounds(const u64 *__restrict__ npay, const u64 *__restrict__ nK,
                                                     const u32 *__restrict__ gseg, const u64 *__restrict__ grow, u64 m, u32 depth,
                                                     SortRec rec, u16 *__restrict__ lcp, u8 *__restrict__ cont, u8 *__restrict__ rhead)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const u64 k = nK[j];
        const u32 sg = gseg[j];
        const bool term = bfq_skey_has_term(k);
        const bool first = (j == 0) || gseg[j - 1] != sg;
        const u64 kp = first ? 0ull : nK[j - 1];
        const bool head = first || kp != k || term;
        const bool nextTied = (j + 1 < m) && gseg[j + 1] == sg && nK[j + 1] == k && !term;
        const u64 r = grow[j];
        rec_set_pay(rec, r, npay[j]);
        if (!first && head) lcp[r] = (u16)(depth + (u32)bfq_skey_lcp(kp, k));
        const bool c = !head || nextTied;
        cont[j] = c ? 1 : 0;
        rhead[j] = (c && head) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void k_huge_compact(const u8 *__restrict__ cont, const u8 *__restrict__ rhead,
                                                      const u64 *__restrict__ cpos, const u64 *__restrict__ hpos,
                                                      const u64 *__restrict__ npay, const u64 *__restrict__ grow, u64 m,
                                                      u64 *__restrict__ grow2, u32 *__restrict__ gseg2, u64 *__restrict__ pay2)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        if (!cont[j]) continue;
        u64 q = cpos[j];
        grow2[q] = grow[j];
        pay2[q] = npay[j];
        gseg2[q] = (u32)(hpos[j] + rhead[j] - 1);      // heads up to and including this run's
    }
}

struct HugeBuf {
    u64 *grow, *grow2, *pay, *K, *npay, *nK, *cpos, *hpos, *hstart, *hoff, *tot;
    u32 *gseg, *gseg2;
    u8 *cont, *rhead;
    SortRec R1, R2;
};
#define HUGE_BYTES_PER_SLOT 112   // the arrays above + radix histograms, rounded up

static void huge_batch(bfq_ctx *c, SortRec rec, const u64 *text3, u16 *lcp, const u64 *hs, const u64 *hl, u32 nseg)
{
    std::vector<u64> off(nseg + 1, 0);
    for (u32 i = 0; i < nseg; i++) off[i + 1] = off[i] + hl[i];
    u64 m = off[nseg];
    size_t mk = c->mark();
    HugeBuf b;
    b.hstart = c->alloc<u64>(nseg); b.hoff = c->alloc<u64>(nseg + 1); b.tot = c->alloc<u64>(2);
    b.grow = c->alloc<u64>(m); b.grow2 = c->alloc<u64>(m); b.pay = c->alloc<u64>(m); b.K = c->alloc<u64>(m);
    b.npay = c->alloc<u64>(m); b.nK = c->alloc<u64>(m); b.cpos = c->alloc<u64>(m); b.hpos = c->alloc<u64>(m);
    b.gseg = c->alloc<u32>(m); b.gseg2 = c->alloc<u32>(m);
    b.cont = c->alloc<u8>(m); b.rhead = c->alloc<u8>(m);
    b.R1.w0 = c->alloc<u32>(m + 16); b.R1.w12 = c->alloc<u64>(m + 16);
    b.R2.w0 = c->alloc<u32>(m + 16); b.R2.w12 = c->alloc<u64>(m + 16);
    HIP_CHECK(hipMemcpyAsync(b.hstart, hs, 8ull * nseg, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(b.hoff, off.data(), 8ull * (nseg + 1), hipMemcpyHostToDevice, c->stream));
    KLAUNCH(c, K_HUGE_ROUND, 32.0 * (double)m, k_huge_expand, bfq_grid(m, 256), 256, (const u64 *)b.hstart, (const u64 *)b.hoff, nseg, m,
            rec, b.grow, b.gseg, b.pay);
    u64 nsub = nseg;                                   // sub-segment ids are dense: 0 .. nsub-1
    for (u32 depth = BFQ_KEY_SYMS; m > 0; depth += BFQ_KEY_SYMS) {
        const unsigned g = bfq_grid(m, 256);
        KLAUNCH(c, K_HUGE_ROUND, 36.0 * (double)m, k_huge_keys, g, 256, (const u64 *)b.pay, text3, m, depth, b.K, b.R1);
        const SortRec S1 = bfq_radix_sort(c, b.R1, b.R2, m);                  // five passes: the result is in R2
        const SortRec T1 = (S1.w0 == b.R1.w0) ? b.R2 : b.R1;
        KLAUNCH(c, K_HUGE_ROUND, 28.0 * (double)m, k_huge_segkeys, g, 256, S1, (const u32 *)b.gseg, m, T1);
        const SortRec S2 = bfq_radix_sort(c, T1, S1, m, nsub <= (1ull << 16) ? 2 : 4);   // ids below 2^16: the two low digits are enough
        KLAUNCH(c, K_HUGE_ROUND, 44.0 * (double)m, k_huge_apply, g, 256, S2, (const u64 *)b.pay, (const u64 *)b.K, m, b.npay, b.nK);
        KLAUNCH(c, K_HUGE_ROUND, 44.0 * (double)m, k_huge_bounds, g, 256, (const u64 *)b.npay, (const u64 *)b.nK, (const u32 *)b.gseg,
                (const u64 *)b.grow, m, depth, rec, lcp, b.cont, b.rhead);
        bfq_exscan_u8(c, b.cont, b.cpos, m, b.tot);
        bfq_exscan_u8(c, b.rhead, b.hpos, m, b.tot + 1);
        KLAUNCH(c, K_HUGE_ROUND, 40.0 * (double)m, k_huge_compact, g, 256, (const u8 *)b.cont, (const u8 *)b.rhead, (const u64 *)b.cpos,
                (const u64 *)b.hpos, (const u64 *)b.npay, (const u64 *)b.grow, m, b.grow2, b.gseg2, b.pay);
        u64 left[2] = {0, 0};                          // slots still tied, sub-segments they form
        HIP_CHECK(hipMemcpyAsync(left, b.tot, 16, hipMemcpyDeviceToHost, c->stream));
        c->sync();
        m = left[0];
        nsub = left[1];
        std::swap(b.grow, b.grow2);
        std::swap(b.gseg, b.gseg2);
        if (depth > BFQ_MAX_READ_LEN + 64) throw BfqError{BFQ_E_TOO_LONG, "suffix comparison ran past the longest read"};
    }
    c->sync();                                         // off[] was copied asynchronously
    c->release(mk);
}

void bfq_refine_bitonic(bfq_ctx *c, SortRec rec, const u64 *text3, u64 n, u16 *lcp, const u64 *d_start, const u64 *d_len, u64 count);

void bfq_refine_huge(bfq_ctx *c, SortRec rec, const u64 *text3, u64 n, u16 *lcp, const u64 *hugeStart, const u64 *hugeLen)
{
    u64 hc[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(hc, &c->d_cnt->hugeCount, 16, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    const u64 nh = hc[0];
    if (!nh) return;
    std::vector<u64> hs(nh), hl(nh);
    HIP_CHECK(hipMemcpyAsync(hs.data(), hugeStart, 8 * nh, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipMemcpyAsync(hl.data(), hugeLen, 8 * nh, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    // slots that fit into what is left of the workspace (the sort's second record buffer is free by now);
    // when the largest segment does not, the rounds run in a side buffer taken from the free device memory
    u64 maxLen = 0;
    for (u64 i = 0; i < nh; i++) maxLen = std::max(maxLen, hl[i]);
    const size_t fixed = (48u << 20) + 16 * nh;
    auto slots = [&](size_t bytes) { return bytes > fixed ? (u64)((bytes - fixed) / HUGE_BYTES_PER_SLOT) : 0ull; };
    u64 cap = slots(c->wsCap - c->wsTop);
    char *side = nullptr;
    char *const ws0 = c->ws;
    const size_t cap0 = c->wsCap, top0 = c->wsTop;
    if (cap < maxLen && !c->env.hugeCap) {
        size_t freeB = 0, totalB = 0;
        HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
        size_t want = fixed + (size_t)std::min<u64>(hc[1], 1ull << 31) * HUGE_BYTES_PER_SLOT;
        size_t most = freeB - freeB / 8;
        if (want > most) want = most;
        if (slots(want) > cap && hipMalloc((void **)&side, want) == hipSuccess) {
            c->ws = side; c->wsCap = want; c->wsTop = 0;
            cap = slots(want);
        } else {
            (void)hipGetLastError();
            side = nullptr;
        }
    }
    struct Restore {                                   // the arena goes back however the rounds end
        bfq_ctx *c; char *ws; size_t cap, top; char *side;
        ~Restore() { if (side) { (void)hipStreamSynchronize(c->stream); (void)hipFree(side); c->ws = ws; c->wsCap = cap; c->wsTop = top; } }
    } restore{c, ws0, cap0, top0, side};
    if (bfq_env().trace) fprintf(stderr, "[bfq huge] %llu segments, %llu rows, longest %llu; slots %llu%s\n", (unsigned long long)nh, (unsigned long long)hc[1],
                                 (unsigned long long)maxLen, (unsigned long long)cap, side ? " (side buffer)" : "");
    if (cap > (1ull << 31)) cap = 1ull << 31;          // slot numbers and sub-segment ids are 32-bit sort keys
    if (c->env.hugeCap && c->env.hugeCap < cap) cap = c->env.hugeCap;   // BFQ_HUGE_CAP, test hook: exercise batching and the oversize route on small inputs
    std::vector<u64> overS, overL;
    u64 i = 0;
    while (i < nh) {
        if (hl[i] > cap) { overS.push_back(hs[i]); overL.push_back(hl[i]); i++; continue; }
        u64 j = i, rows = 0;
        while (j < nh && hl[j] <= cap - rows && j - i < (1u << 30)) rows += hl[j++];
        huge_batch(c, rec, text3, lcp, hs.data() + i, hl.data() + i, (u32)(j - i));
        i = j;
    }
    if (!overS.empty()) {                              // larger than the free workspace: one workgroup each after all
        size_t mk = c->mark();
        u64 *ds = c->alloc<u64>(overS.size()), *dl = c->alloc<u64>(overS.size());
        HIP_CHECK(hipMemcpyAsync(ds, overS.data(), 8 * overS.size(), hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(dl, overL.data(), 8 * overL.size(), hipMemcpyHostToDevice, c->stream));
        bfq_refine_bitonic(c, rec, text3, n, lcp, ds, dl, overS.size());
        c->sync();
        c->release(mk);
    }
}
