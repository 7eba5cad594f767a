// k_refine.hip -- finishes the suffix order after the radix sort of the key prefix and
// produces the LCP array, then the eBWT / permuted-quality bytes.
//
// After the sort, rows equal on the sorted prefix (the first keySyms symbols) form
// segments.  A prefix that holds a terminator is a complete suffix: equal such keys are
// identical suffixes, already in read order (stable sort, #_i < #_j for i < j), so they
// are final.  All other segments of >= 2 rows are refined on the following 21-symbol
// words of the text:
//   k_refine_chunk : one WAVEFRONT per chunk of 2048 rows, no workgroup barriers.  Segment heads (and the LCP of
//                    head rows) come straight from the sorted keys (4 rows per lane and step, head bits OR-ed
//                    into wave-private LDS words); the chunk's segments of >= 2 rows are compacted into LDS;
//                    then up to 64 rows of several segments are packed one row per lane and refined together,
//                    round by round (42 symbols each), entirely in registers:
//                      next two 21-symbol words of every open row (fetched by word index, no division),
//                      stable rank inside each sub-segment by one-directional compares over lane offsets
//                      (DPP wave shifts) + ballot complement, one ds_permute of the payload and the words,
//                      new sub-segment heads and their LCP from the words.
//   k_refine_big   : segments of 65 .. BFQ_HUGE_SEG rows: one workgroup (or one wavefront) each, everything in
//                    LDS -- a bitonic network over (sub-segment id, two words, original slot) that moves only a
//                    4-byte (id, slot) word; five size classes by LDS footprint.
//                    Segments above BFQ_HUGE_SEG rows (low-complexity reads: poly-A/G tails, short tandem
//                    repeats) are only listed here and sorted by whole-device radix rounds (k_bigseg.hip);
//                    k_refine_listed (a global-memory bitonic network, one workgroup per segment) is the last
//                    resort for segments beyond even those rounds' workspace.
// LCP convention: common prefix counted on bases only, terminators never match
// (what bfq_int deduces from the BWT, bfq_int.cpp:139-181,183-300, and what
// eGap --lcp hands to bfq_ext, bfq_ext.cpp:350-412).
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_rec.h"

#define LCP_PENDING 0xFFFFu
#define RF_CHUNK 2048                               // rows per wavefront chunk
#define RF_HBW (RF_CHUNK / 64 + 4)                  // head-bit words per chunk: its rows + 256 rows of lookahead

// payload <-> working form: word index (34 bits) | 3*offset (6 bits) << 34 | code << 40 | quality << 43 of p + 16
__device__ __forceinline__ u64 wo_from_pay(u64 pay)
{
    u64 p = bfq_val_pos(pay) + BFQ_KEY_SYMS;
    u64 w = p / BFQ_SYMS_PER_WORD;
    u64 o3 = (p - w * BFQ_SYMS_PER_WORD) * 3ull;
    return w | (o3 << 34) | ((u64)bfq_val_code(pay) << 40) | ((u64)bfq_val_qual(pay) << 43);
}
__device__ __forceinline__ u64 wo_to_pay(u64 v)
{
    u64 w = v & ((1ull << 34) - 1ull), o3 = (v >> 34) & 63ull;
    u64 p = w * BFQ_SYMS_PER_WORD + o3 / 3ull - BFQ_KEY_SYMS;
    return bfq_pack_val(p, (u32)(v >> 40) & 7u, (u32)(v >> 43) & 0xFFu);
}
// the two consecutive masked windows `round` and `round + 1` words along the suffix (three text words);
// the second one is 0 when the first already holds the terminator
__device__ __forceinline__ void wo_key2(const u64 *__restrict__ text3, u64 v, u32 round, u64 &W1, u64 &W2)
{
    u64 w = (v & ((1ull << 34) - 1ull)) + round;
    u32 o = (u32)(v >> 34) & 63u;
    // two memory requests instead of three: one 16-byte load for the first two words (8-byte alignment is enough for it);
    // same-box A/B at 30 M x 150: k_refine_chunk 146.5 -> 140.0 ms
    typedef unsigned long long u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));   // the address is 8-byte aligned only: say so (ulonglong2 claims 16)
    const u64 *t3 = text3 + bfq_t3_at(w);                        // the three words lie in one 64-byte sector
    const u64x2_a8 t01 = *(const u64x2_a8 *)t3;
    const u64 t0 = t01.x, t1 = t01.y, t2 = t3[2];
    u64 a = (t0 << o) & BFQ_M63, b = (t1 << o) & BFQ_M63;
    if (o) { a |= t1 >> (63u - o); b |= t2 >> (63u - o); }
    W1 = bfq_mask_key(a);
    W2 = bfq_key_has_term(W1) ? 0ull : bfq_mask_key(b);
}

__global__ __launch_bounds__(256) void k_refine_chunk(SortRec rec, u16 *__restrict__ lcp, const u64 *__restrict__ text3,
                                                      u64 n, u64 *__restrict__ biglist, u64 *__restrict__ biglen, DevCounters *cnt,
                                                      u64 nchunks, u16 *__restrict__ firstHead)
{
    // every wavefront works alone on its own chunk (no workgroup barriers): private LDS slices
    __shared__ u64 hb_all[4][RF_HBW];               // head bits of rows [base, base + RF_CHUNK + 256)
    __shared__ u32 segs_all[4][RF_CHUNK / 2];       // chunk-local start | size << 16  (size 0: longer than a wavefront)
    const u32 lane = bfq_lane(), w = threadIdx.x >> 6;
    u64 *hb = hb_all[w];
    u32 *segs = segs_all[w];
    const u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 ch = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; ch < nchunks; ch += nwaves) {
        const u64 base = ch * RF_CHUNK;
        // 1. segment heads from the sorted keys (+ LCP of this chunk's head rows): 4 rows per lane and step
        //    (16 B of w0, 32 B of w12), head bits OR-ed into LDS words in row order
        for (u32 i = lane; i < RF_HBW; i += 64) hb[i] = 0;
        bfq_wave_sync();
#pragma unroll 1
        for (u32 g = 0; g < RF_HBW / 4; g++) {
            const u32 li0 = g * 256 + lane * 4;
            const u64 r0 = base + li0;
            u64 k[4];
            if (li0 >= RF_CHUNK + 128) {                       // beyond the lookahead any segment search needs
                k[0] = k[1] = k[2] = k[3] = 0ull;
            } else if (r0 + 4 <= n) {
                uint4 a = *(const uint4 *)(rec.w0 + r0);
                uint4 b0 = *(const uint4 *)(rec.w12 + r0), b1 = *(const uint4 *)(rec.w12 + r0 + 2);
                k[0] = bfq_rec_skey(a.x, b0.x); k[1] = bfq_rec_skey(a.y, b0.z);
                k[2] = bfq_rec_skey(a.z, b1.x); k[3] = bfq_rec_skey(a.w, b1.z);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) k[i] = (r0 + i < n) ? rec_key(rec, r0 + i) : 0ull;
            }
            u64 kp = bfq_from_prev_lane(k[3]);                 // key of row r0-1: the previous lane's last, one extra load for lane 0
            if (lane == 0 && r0 && r0 - 1 < n) kp = rec_key(rec, r0 - 1);
            u32 nib = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u64 r = r0 + i;
                bool h = true;                                 // rows past the end close the last segment
                if (r < n) {
                    h = (r == 0) || seg_head(kp, k[i]);
                    if (h && li0 + i < RF_CHUNK) lcp[r] = r ? (u16)bfq_skey_lcp(kp, k[i]) : (u16)0;
                }
                nib |= (h ? 1u : 0u) << i;
                kp = k[i];
            }
            atomicOr((unsigned long long *)&hb[g * 4 + (lane >> 4)], (unsigned long long)nib << (4 * (lane & 15)));
        }
        bfq_wave_sync();
        // 2. the chunk's segments of >= 2 rows, in row order, into LDS: one head word (64 rows) per lane
        u32 nsegs;
        {
            const bool on = lane < RF_CHUNK / 64;
            const u64 hw = on ? hb[lane] : ~0ull, hn = on ? hb[lane + 1] : ~0ull;
            {                                                      // first head row of the chunk (RF_CHUNK: none), for k_refine_big's search
                const u64 any = __ballot(on && hw != 0);
                const int fl = any ? __builtin_ctzll(any) : -1;
                const u64 fw = bfq_readlane64(hw, fl < 0 ? 0 : fl);
                if (lane == 0) firstHead[ch] = (u16)(fl < 0 ? RF_CHUNK : fl * 64 + __builtin_ctzll(fw));
            }
            u64 S = on ? (hw & ~((hw >> 1) | (hn << 63))) : 0ull;   // head followed by a non-head (rows past n are heads)
            const u32 c = (u32)__popcll(S);
            const u32 incl = bfq_wave_incscan32(c);
            u32 o = incl - c;
            while (S) {
                const u32 li = lane * 64 + (u32)__builtin_ctzll(S);
                S &= S - 1;
                const u32 idx = li + 1;                            // first row that may be the next head
                const u64 wd = hb[idx >> 6] >> (idx & 63);
                u32 nxt;
                if (wd) nxt = idx + (u32)__builtin_ctzll(wd);
                else {                                             // head bits are valid up to the lookahead (RF_CHUNK + 128 rows)
                    nxt = 0xFFFFu;
                    for (u32 wi = (idx >> 6) + 1; wi < RF_CHUNK / 64 + 2; wi++) {
                        const u64 w2 = hb[wi];
                        if (w2) { nxt = (wi << 6) + (u32)__builtin_ctzll(w2); break; }
                    }
                }
                const u32 size = (nxt == 0xFFFFu) ? 0xFFFFu : nxt - li;      // 0xFFFF: ends beyond the lookahead
                segs[o++] = li | (size << 16);                     // more than 64 rows: handled by k_refine_big
            }
            nsegs = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        }
        bfq_wave_sync();
        if (lane == 0 && nsegs) atomicAdd(&cnt->nSegs, (u64)nsegs);
        // 3. refinement: each wavefront takes 64 segments at a time
        for (u32 b0 = 0; b0 < nsegs; b0 += 64) {
            u32 idx = b0 + lane;
            u32 ent = (idx < nsegs) ? segs[idx] : 0u;
            u64 segStart = base + (ent & 0xFFFFu);
            u32 segSize = (idx < nsegs) ? (ent >> 16) : 0u;
            if (segSize > 64) {                                    // listed with its length when that is known (0: to be searched)
                const u64 k = atomicAdd(&cnt->bigCount, 1ull);
                biglist[k] = segStart;
                biglen[k] = (segSize == 0xFFFFu) ? 0ull : (u64)segSize;
                segSize = 0;
            }
            u32 incl = bfq_wave_incscan32(segSize);
            u32 excl = incl - segSize;
            u32 done = 0;
            while (done < 64) {
                u32 sbase = (u32)__builtin_amdgcn_readlane((int)excl, (int)done);
                u64 take = __ballot(lane >= done && incl - sbase <= 64u);
                u32 ntake = (u32)__popcll(take);                                    // >= 1
                u32 rows = (u32)__builtin_amdgcn_readlane((int)incl, (int)(done + ntake - 1)) - sbase;
                if (rows == 0) { done += ntake; continue; }
                // rows -> lanes
                u64 myRow = 0;
                int sublo = 0, subhi = 0, seglo = 0;
                for (u32 t = 0; t < ntake; t++) {
                    int k = (int)(done + t);
                    u32 sz = (u32)__builtin_amdgcn_readlane((int)segSize, k);
                    if (!sz) continue;                                               // uniform
                    int hp = (int)((u32)__builtin_amdgcn_readlane((int)excl, k) - sbase);
                    u64 st = bfq_readlane64(segStart, k);
                    if ((int)lane >= hp && (int)lane < hp + (int)sz) { myRow = st + (u64)((int)lane - hp); sublo = hp; subhi = hp + (int)sz; seglo = hp; }
                }
                const bool act = lane < rows;
                // working form of the payload: word index and bit offset of text position p + 16 instead of p,
                // so that the word of every later round (+21 symbols = +1 word) needs no division
                // the key bits of a segment's rows are all the same: kept from this load, the write-back needs no second one
                const u64 x0 = act ? rec.w12[myRow] : 0ull;
                const u32 keyBits = (u32)x0 & 0xFFFF0000u;
                u64 v = act ? wo_from_pay(bfq_rec_pay((u32)x0, (u32)(x0 >> 32))) : 0ull;
                u32 mylcp = LCP_PENDING;                                            // positional: LCP(row-1,row)
                u64 unres = __ballot(act);
                u64 curHeads = ~0ull;                                               // head lanes of the current sub-segments
                curHeads = __ballot(!act || (int)lane == sublo);
                u32 depth = BFQ_KEY_SYMS;
                u32 round = 0;
                while (unres) {                                                     // one round = the next 42 symbols
                    const bool un = (unres >> lane) & 1ull;
                    u64 W1 = 0, W2 = 0;
                    if (un) wo_key2(text3, v, round, W1, W2);
                    // longest open sub-segment = longest run of non-head lanes + 1 (scalar bit trick on the head mask)
                    u32 maxsz = 1;
                    for (u64 run = ~curHeads; run; run &= run >> 1) maxsz++;
                    // stable rank of a row inside its sub-segment = #(later rows with a smaller word pair)
                    // + #(earlier rows with a pair <= its own).  Only the first kind is compared
                    // (the pair of lane+d arrives by one-lane DPP shifts per step); the ballot of those
                    // comparisons, read at lane-d, gives the second kind by complement.
                    // Which lanes take part is wave-uniform: lane l and lane l+d share a sub-segment iff lanes l+1 .. l+d are
                    // no heads (scalar mask arithmetic); per step the vector unit only shifts, compares and counts.
                    int c = (int)lane - sublo;                                       // (rows before me in my sub-segment) - inv + (later rows before me)
                    u64 U1 = W1, U2 = W2;
                    u64 same = ~0ull;
                    const u64 nh = ~curHeads;
                    for (u32 d = 1; d < maxsz; d++) {
                        same &= nh >> d;
                        U1 = bfq_from_next_lane(U1);
                        U2 = bfq_from_next_lane(U2);
                        const u64 fb = __builtin_amdgcn_ballot_w64(U1 < W1 || (U1 == W1 && U2 < W2)) & same;   // bit l: row l+d sorts before row l
                        c += __builtin_amdgcn_inverse_ballot_w64(fb) ? 1 : 0;
                        c -= __builtin_amdgcn_inverse_ballot_w64(fb << d) ? 1 : 0;            // I sort before row lane-d
                    }
                    int np = un ? sublo + c : (int)lane;
                    v = bfq_permute64(v, np);
                    W1 = bfq_permute64(W1, np);
                    W2 = bfq_permute64(W2, np);
                    const u64 P1 = bfq_from_prev_lane(W1), P2 = bfq_from_prev_lane(W2);
                    const bool d1 = (W1 != P1) || bfq_key_has_term(W1);             // decided by the first word
                    bool newhead = un && ((int)lane == sublo || d1 || W2 != P2 || bfq_key_has_term(W2));
                    if (un && (int)lane != sublo && newhead)
                        mylcp = d1 ? depth + (u32)bfq_key_lcp(P1, W1) : depth + BFQ_SYMS_PER_WORD + (u32)bfq_key_lcp(P2, W2);
                    u64 heads = __ballot(newhead || !un);
                    curHeads = heads;
                    sublo = 63 - __clzll((long long)(heads & le));
                    u64 above = heads & ~le;
                    subhi = above ? __builtin_ctzll(above) : 64;
                    unres = __ballot(un && (subhi - sublo > 1));
                    depth += 2 * BFQ_SYMS_PER_WORD;
                    round += 2;
                }
                if (act) {
                    const u64 pay = wo_to_pay(v);
                    rec.w12[myRow] = ((u64)(u32)pay << 32) | keyBits | (u32)(pay >> 32);
                    if ((int)lane != seglo) lcp[myRow] = (u16)mylcp;
                }
                done += ntake;
            }
        }
        bfq_wave_sync();                       // hb / segs are reused by the next chunk
    }
}

// ---- larger segments: one workgroup each, bitonic network in global memory -------
// full-suffix order beyond the sorted prefix; ties (identical suffixes) by position
__device__ bool suffix_less(const u64 *__restrict__ text3, u64 pa, u64 pb)
{
    for (u32 d = BFQ_KEY_SYMS;; d += BFQ_SYMS_PER_WORD) {
        u64 a = bfq_key_at(text3, pa + d), b = bfq_key_at(text3, pb + d);
        if (a != b) return a < b;
        if (bfq_key_has_term(a)) return pa < pb;
    }
}
__device__ u32 suffix_lcp(const u64 *__restrict__ text3, u64 pa, u64 pb)
{
    for (u32 d = BFQ_KEY_SYMS;; d += BFQ_SYMS_PER_WORD) {
        u64 a = bfq_key_at(text3, pa + d), b = bfq_key_at(text3, pb + d);
        if (a != b || bfq_key_has_term(a)) return d + (u32)bfq_key_lcp(a, b);
    }
}

__device__ __forceinline__ void big_step(const SortRec &rec, u64 s, u64 g, u64 j, const u64 *__restrict__ text3)
{
    for (u64 i = threadIdx.x; i < g; i += 256) {
        u64 l = i ^ j;
        if (l > i && l < g) {                      // rows >= g are a virtual +inf padding
            u64 va = rec_pay(rec, s + i), vb = rec_pay(rec, s + l);
            if (suffix_less(text3, bfq_val_pos(vb), bfq_val_pos(va))) { rec_set_pay(rec, s + i, vb); rec_set_pay(rec, s + l, va); }
        }
    }
    __syncthreads();
}

// bitonic network with ascending comparators only: flip (i <-> i^(k-1)) then
// disperse (i <-> i^j, j = k/4 .. 1); the +inf padding beyond g never moves
__device__ void big_sort(const SortRec &rec, u64 s, u64 g, const u64 *__restrict__ text3, u16 *__restrict__ lcp)
{
    u64 P = 1;
    while (P < g) P <<= 1;
    for (u64 k = 2; k <= P; k <<= 1) {
        big_step(rec, s, g, k - 1, text3);
        for (u64 j = k >> 2; j >= 1; j >>= 1) big_step(rec, s, g, j, text3);
    }
    for (u64 i = 1 + threadIdx.x; i < g; i += 256)
        lcp[s + i] = (u16)suffix_lcp(text3, bfq_val_pos(rec_pay(rec, s + i - 1)), bfq_val_pos(rec_pay(rec, s + i)));
}

// ---- segments of 65 .. BFQ_HUGE_SEG rows: one workgroup each, everything in LDS ------------------------
// Round by round (42 symbols each) like the wavefront kernel, but the stable rank comes from a bitonic network
// over (sub-segment id, next word, original slot) kept in LDS: every row's next word is fetched once per
// round, no suffix is compared through global memory.  Slot i of the segment is eBWT row s + i throughout, so
// the LCP of a boundary is final the moment the boundary appears.
// Five size classes, set by their LDS footprint (= how many run per CU): one wavefront (64 threads, no real barriers) for
// segments up to 128, 256 and RB_SMALL rows, 128 threads up to twice that, 256 threads up to BFQ_HUGE_SEG (one wavefront
// for the larger classes measured slower: LDS latency).
#define RB_SMALL 512
template <int PMAX> struct BlockSortLds {
    u64 W1[PMAX], W2[PMAX];   // by ORIGINAL slot: the row's next two words (0 for rows that are already alone)
    u32 T[PMAX];          // by current slot: sub-segment id << 12 | original slot (padding: all ones); all the network moves
    u64 V[PMAX];          // by ORIGINAL slot: working form of the payload (word index / offset of p + 16)
    u32 scan[4];
    u32 open;
};
template <int NT> __device__ __forceinline__ void grp_sync()
{
    if (NT == 64) {
        // one wavefront: lanes exchange data through LDS across this point, so the scheduling barrier is fenced --
        // the compiler must neither forward a lane's own LDS value nor hoist LDS accesses over it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else __syncthreads();
}
template <int NT, int PMAX>
__device__ void block_sort(const SortRec &rec, u64 s, u32 g, const u64 *__restrict__ text3, u16 *__restrict__ lcp, BlockSortLds<PMAX> &L)
{
    const u32 tid = threadIdx.x;
    u32 P = 128;
    while (P < g) P <<= 1;
    for (u32 i = tid; i < P; i += NT) {
        if (i < g) { L.V[i] = wo_from_pay(rec_pay(rec, s + i)); L.T[i] = i; }
        else L.T[i] = 0xFFFFFFFFu;
    }
    grp_sync<NT>();
    u32 depth = BFQ_KEY_SYMS;
    for (u32 round = 0;; round += 2, depth += 2 * BFQ_SYMS_PER_WORD) {        // one round = the next 42 symbols
        // next words of every row that still shares its sub-segment with a neighbour
        for (u32 i = tid; i < g; i += NT) {
            const u32 t = L.T[i], sub = t >> 12;
            const bool alone = (i == 0 || (L.T[i - 1] >> 12) != sub) && (i + 1 >= g || (L.T[i + 1] >> 12) != sub);
            u64 w1 = 0, w2 = 0;
            if (!alone) wo_key2(text3, L.V[t & 0xFFFu], round, w1, w2);
            L.W1[t & 0xFFFu] = w1; L.W2[t & 0xFFFu] = w2;          // words stay with the original slot: the network moves T only
        }
        grp_sync<NT>();
        // bitonic network on (sub-segment, words, original slot); padding sorts last
        for (u32 k = 2; k <= P; k <<= 1)
            for (u32 j = k >> 1; j >= 1; j >>= 1) {
                for (u32 c = tid; c < P / 2; c += NT) {
                    const u32 lo = ((c & ~(j - 1)) << 1) | (c & (j - 1)), hi = lo | j;
                    const bool up = (lo & k) == 0;
                    const u32 ta = L.T[lo], tb = L.T[hi];
                    bool less;                                     // (sub-segment, words, original slot) of hi < that of lo
                    if ((ta >> 12) != (tb >> 12)) less = (tb >> 12) < (ta >> 12);
                    else if (tb == 0xFFFFFFFFu) less = false;      // two padding slots
                    else {
                        const u32 ia = ta & 0xFFFu, ib = tb & 0xFFFu;
                        const u64 a1 = L.W1[ia], b1 = L.W1[ib];
                        if (a1 != b1) less = b1 < a1;
                        else { const u64 a2 = L.W2[ia], b2 = L.W2[ib]; less = (a2 != b2) ? (b2 < a2) : (ib < ia); }
                    }
                    if (less == up) { L.T[lo] = tb; L.T[hi] = ta; }
                }
                grp_sync<NT>();
            }
        // new boundaries, their LCP, new dense sub-segment ids (block scan of the head flags, 8 slots per thread at most)
        u32 heads = 0, cnt = 0;
        const u32 per = P / NT ? P / NT : 1, i0 = tid * per;
        for (u32 q = 0; q < per; q++) {
            const u32 i = i0 + q;
            if (i >= g || (P < NT && tid >= P)) break;
            bool h = true;
            if (i) {
                const u32 t = L.T[i], tp = L.T[i - 1];
                const u64 w1 = L.W1[t & 0xFFFu], p1 = L.W1[tp & 0xFFFu], w2 = L.W2[t & 0xFFFu], p2 = L.W2[tp & 0xFFFu];
                const bool sameSub = (t >> 12) == (tp >> 12);
                const bool d1 = w1 != p1 || bfq_key_has_term(w1);                  // decided by the first word
                h = !sameSub || d1 || w2 != p2 || bfq_key_has_term(w2);
                if (sameSub && h)
                    lcp[s + i] = (u16)(d1 ? depth + (u32)bfq_key_lcp(p1, w1) : depth + BFQ_SYMS_PER_WORD + (u32)bfq_key_lcp(p2, w2));
            }
            heads |= (h ? 1u : 0u) << q;
            cnt += h ? 1u : 0u;
        }
        u32 ex = bfq_wave_incscan32(cnt) - cnt;                     // exclusive scan over the NT threads
        if (NT > 64) {
            const u32 w = tid >> 6;
            if ((tid & 63u) == 63u) L.scan[w] = ex + cnt;
            __syncthreads();
            for (u32 k = 0; k < w; k++) ex += L.scan[k];
        }
        grp_sync<NT>();                                             // all reads of T above are done
        if (tid == 0) L.open = 0;
        grp_sync<NT>();
        u32 id = ex;                                                // heads before my first slot
        bool anyOpen = false;
        for (u32 q = 0; q < per; q++) {
            const u32 i = i0 + q;
            if (i >= g || (P < NT && tid >= P)) break;
            if ((heads >> q) & 1u) id++;
            else anyOpen = true;                                    // a slot that is not a head shares its sub-segment
            L.T[i] = ((id - 1) << 12) | (L.T[i] & 0xFFFu);
        }
        if (anyOpen) L.open = 1;
        grp_sync<NT>();
        if (!L.open) break;                                         // uniform
    }
    // final order: slot i takes the payload of original slot T[i] & 0xFFF (all reads before any write)
    u64 pay[PMAX / NT];
    for (u32 q = 0, i = tid; i < g; i += NT, q++) pay[q] = rec_pay(rec, s + (L.T[i] & 0xFFFu));
    grp_sync<NT>();
    for (u32 q = 0, i = tid; i < g; i += NT, q++) rec_set_pay(rec, s + i, pay[q]);
    grp_sync<NT>();
}

// One workgroup of NT threads per listed segment.  The 64-thread launch comes first: it finds every segment's
// extent (kept in biglen) and sorts those of up to RB_SMALL rows; the 128- and 256-thread launches take the next
// two size classes (their LDS footprint sets how many run per CU), the last one lists what is longer than
// BFQ_HUGE_SEG rows for the radix rounds.
template <int NT, int PMIN, int PMAX, bool FIRST>
__global__ __launch_bounds__(NT) void k_refine_big(const u64 *__restrict__ biglist, u64 *__restrict__ biglen, DevCounters *cnt, SortRec rec,
                                                   u16 *__restrict__ lcp, const u64 *__restrict__ text3, u64 n,
                                                   u64 *__restrict__ hugeStart, u64 *__restrict__ hugeLen,
                                                   const u16 *__restrict__ firstHead)
{
    __shared__ u64 shEnd;
    __shared__ BlockSortLds<PMAX> L;
    const u64 nbig = cnt->bigCount;
    for (u64 bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
        const u64 s = biglist[bi];
        u64 g;
        if (FIRST && (g = biglen[bi]) != 0) {
            if (g > PMAX) continue;                    // uniform; length known from the chunk kernel
            grp_sync<NT>();
        } else if (FIRST) {
            grp_sync<NT>();
            if (threadIdx.x == 0) shEnd = ~0ull;
            grp_sync<NT>();
            // first head after s = end of the segment: the rest of s's chunk 512 rows at a time, then whole chunks
            // by their first-head entries
            const u64 cend = (s / RF_CHUNK + 1) * RF_CHUNK;
            for (u64 b0 = s + 1; b0 < cend; b0 += NT * 8) {
                u64 i0 = b0 + (u64)threadIdx.x * 8;
                u64 kp = (i0 < n && i0 < cend) ? rec_key(rec, i0 - 1) : 0ull;
                for (u32 k = 0; k < 8; k++) {
                    u64 i = i0 + k;
                    if (i >= cend) break;
                    if (i >= n) { atomicMin(&shEnd, i); break; }
                    u64 kc = rec_key(rec, i);
                    if (seg_head(kp, kc)) { atomicMin(&shEnd, i); break; }
                    kp = kc;
                }
                grp_sync<NT>();
                if (shEnd != ~0ull) break;             // uniform
            }
            grp_sync<NT>();
            const u64 nch = (n + RF_CHUNK - 1) / RF_CHUNK;
            for (u64 c0 = s / RF_CHUNK + 1;; c0 += NT) {
                bool done = (shEnd != ~0ull);
                grp_sync<NT>();
                if (done) break;                       // uniform
                u64 ch = c0 + threadIdx.x;
                if (ch >= nch) atomicMin(&shEnd, n);
                else { u32 fh = firstHead[ch]; if (fh < RF_CHUNK) atomicMin(&shEnd, ch * RF_CHUNK + fh); }
                grp_sync<NT>();
            }
            g = shEnd - s;
            if (threadIdx.x == 0) biglen[bi] = g;
            if (g > PMAX) continue;                    // uniform
        } else {
            g = biglen[bi];
            if (g <= PMIN) continue;                   // done by a smaller launch
            if (g > BFQ_HUGE_SEG) {                    // uniform: left to the radix rounds
                if (PMAX == BFQ_HUGE_SEG && threadIdx.x == 0) {
                    u64 h = atomicAdd(&cnt->hugeCount, 1ull);
                    atomicAdd(&cnt->hugeRows, g);
                    hugeStart[h] = s; hugeLen[h] = g;
                }
                continue;
            }
            if (g > PMAX) continue;                    // a larger launch's
            grp_sync<NT>();
        }
        block_sort<NT, PMAX>(rec, s, (u32)g, text3, lcp, L);
    }
}

// the global-memory network for listed segments of known length (huge segments that did not fit the radix rounds' workspace)
__global__ __launch_bounds__(256) void k_refine_listed(const u64 *__restrict__ start, const u64 *__restrict__ len, u64 count, SortRec rec,
                                                       u16 *__restrict__ lcp, const u64 *__restrict__ text3)
{
    for (u64 bi = blockIdx.x; bi < count; bi += gridDim.x) {
        __syncthreads();
        big_sort(rec, start[bi], len[bi], text3, lcp);
    }
}

// eBWT byte and permuted quality of every row, from the sort payload
// One wavefront per group of 256 rows, 4 rows per lane (32 contiguous bytes of payload in, 4 + 4 bytes out);
// the group's symbol counts (what k_lf_count would recount from the eBWT bytes) come out on the way.
__global__ __launch_bounds__(256) void k_emit_bwt(SortRec rec, u64 n, u32 termOut, u8 *__restrict__ bwt, u8 *__restrict__ qs,
                                                  u32 *__restrict__ gcnt, u64 ngroups)
{
    const u32 lane = bfq_lane();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; g < ngroups; g += nwaves) {
        const u64 r0 = g * 256 + (u64)lane * 4;
        u32 b4 = 0, q4 = 0;
        u64 p = 0;
        u64 x[4] = {0, 0, 0, 0};
        if (r0 + 4 <= n) {
            uint4 a = *(const uint4 *)(rec.w12 + r0), b = *(const uint4 *)(rec.w12 + r0 + 2);
            x[0] = ((u64)a.y << 32) | a.x; x[1] = ((u64)a.w << 32) | a.z;
            x[2] = ((u64)b.y << 32) | b.x; x[3] = ((u64)b.w << 32) | b.z;
        } else {
            for (u64 k = 0; r0 + k < n; k++) x[k] = rec.w12[r0 + k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (r0 + k < n) {
                u64 v = bfq_rec_pay((u32)x[k], (u32)(x[k] >> 32));
                u32 code = bfq_val_code(v);
                b4 |= (u32)(code ? bfq_code_sym(code) : (u8)termOut) << (8 * k);
                q4 |= bfq_val_qual(v) << (8 * k);
                p += 1ull << (10 * code);
            }
        }
        if (r0 + 4 <= n) { *(u32 *)(bwt + r0) = b4; *(u32 *)(qs + r0) = q4; }
        else for (u64 k = 0; r0 + k < n; k++) { bwt[r0 + k] = (u8)(b4 >> (8 * k)); qs[r0 + k] = (u8)(q4 >> (8 * k)); }
        p = bfq_readlane64(bfq_wave_incscan64(p), 63);         // the group's packed symbol counts
        if (gcnt && lane < 6) gcnt[(u64)lane * ngroups + g] = (u32)(p >> (10 * lane)) & 0x3FFu;
    }
}

void bfq_refine(bfq_ctx *c, SortRec rec, const u64 *text3, u64 n, u16 *lcp, bfq_stats *st)
{
    (void)st;
    if (!n) return;
    size_t m = c->mark();
    u64 *biglist = c->alloc<u64>(n / 65 + 2);
    u64 nchunks = ceil_div(n, RF_CHUNK);
    u16 *firstHead = c->alloc<u16>(nchunks + 1);
    // remaining packed suffix read once (3(L+1)/16 B) + order 8 B + LCP 1 B per row
    const double lavg = c->N ? (double)(n - c->N) / (double)c->N : 0.0;
    u64 *biglen = c->alloc<u64>(n / 65 + 2);
    KLAUNCH(c, K_REFINE_WAVE, (3.0 * (lavg + 1.0) / 16.0 + 9.0) * (double)n, k_refine_chunk, bfq_grid(nchunks, 4), 256, rec, lcp, text3, n, biglist,
            biglen, c->d_cnt, nchunks, firstHead);
    u64 *hugeStart = c->alloc<u64>(n / BFQ_HUGE_SEG + 2), *hugeLen = c->alloc<u64>(n / BFQ_HUGE_SEG + 2);
    // the list length stays on the device: a fixed grid strides over it (usually empty)
#define RB_LAUNCH(NT, PMIN, PMAX, FIRST, GRID)                                                                              \
    KLAUNCH(c, K_REFINE_BIG, 0.0, (k_refine_big<NT, PMIN, PMAX, FIRST>), GRID, NT, (const u64 *)biglist, biglen, c->d_cnt, rec, lcp, text3, n, \
            hugeStart, hugeLen, (const u16 *)firstHead)
    RB_LAUNCH(64, 64, 128, true, 8192);
    RB_LAUNCH(64, 128, 256, false, 8192);
    RB_LAUNCH(64, 256, RB_SMALL, false, 4096);
    RB_LAUNCH(128, RB_SMALL, 2 * RB_SMALL, false, 2048);
    RB_LAUNCH(256, 2 * RB_SMALL, BFQ_HUGE_SEG, false, 1024);
#undef RB_LAUNCH
    bfq_refine_huge(c, rec, text3, n, lcp, hugeStart, hugeLen);
    c->release(m);
}

void bfq_refine_bitonic(bfq_ctx *c, SortRec rec, const u64 *text3, u64 n, u16 *lcp, const u64 *d_start, const u64 *d_len, u64 count)
{
    (void)n;
    if (!count) return;
    KLAUNCH(c, K_REFINE_BIG, 0.0, k_refine_listed, (unsigned)(count < 1024 ? count : 1024), 256, d_start, d_len, count, rec, lcp, text3);
}

void bfq_emit_bwt(bfq_ctx *c, SortRec rec, u64 n, int termOut, u8 *bwt, u8 *qs, u32 *gcnt)
{
    if (!n) return;
    u64 ngroups = n / 256 + 1;
    KLAUNCH(c, K_EMIT, 10.0 * (double)n, k_emit_bwt, bfq_grid(ngroups, 4), 256, rec, n, (u32)(termOut & 0xFF), bwt, qs, gcnt, ngroups);
}
