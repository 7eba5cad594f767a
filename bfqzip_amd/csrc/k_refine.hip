// k_refine.hip -- finishes the suffix order after the 21-symbol radix sort and
// produces the LCP array, then the eBWT / permuted-quality bytes.
//
// After the sort, equal keys form segments.  A key that holds a terminator is a
// complete suffix: equal such keys are identical suffixes, already in read order
// (stable sort, #_i < #_j for i < j), so they are final.  All other segments of
// >= 2 rows are refined on the following 21-symbol words of the text:
//   k_refine_wave : segments of <= 64 rows, several packed into one wavefront (one row per
//                   lane), all rounds in registers (stable rank by counting over lane
//                   offsets + ds_permute), next word prefetched, LCP as a by-product
//   k_refine_big  : one workgroup per larger segment, bitonic network over the
//                   rows in global memory with a full suffix comparator
// LCP convention: common prefix counted on bases only, terminators never match
// (what bfq_int deduces from the BWT, bfq_int.cpp:139-181,183-300, and what
// eGap --lcp hands to bfq_ext, bfq_ext.cpp:350-412).
#include "bfq_internal.h"
#include "bfq_device.h"

#define LCP_PENDING 0xFFFFu

// head[r] = 1 if row r starts a segment; LCP of head rows comes from the keys
__global__ __launch_bounds__(256) void k_seg_flags(const u64 *__restrict__ keys, u64 n, int lowbit, u8 *__restrict__ head,
                                                   u16 *__restrict__ lcp)
{
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x) {
        u64 k = keys[r];
        if (r == 0) { head[0] = 1; lcp[0] = 0; continue; }
        u64 kp = keys[r - 1];
        // only the radix-sorted prefix (bits >= lowbit) delimits segments; a terminator inside it = complete suffix
        bool h = (((k ^ kp) >> lowbit) != 0) || ((bfq_zero_fields(k) >> lowbit) != 0);
        head[r] = h ? 1 : 0;
        lcp[r] = h ? (u16)bfq_key_lcp(kp, k) : (u16)LCP_PENDING;
    }
}

// segment starts: head[r] && !head[r+1]  (segments of >= 2 rows)
#define SG_CHUNK 4096
__device__ __forceinline__ bool seg_start(const u8 *head, u64 r, u64 n) { return r + 1 < n && head[r] && !head[r + 1]; }

__global__ __launch_bounds__(256) void k_seg_count(const u8 *__restrict__ head, u64 n, u32 *__restrict__ counts)
{
    __shared__ u32 sh[4];
    u64 base = (u64)blockIdx.x * SG_CHUNK;
    u32 c = 0;
    for (int k = 0; k < SG_CHUNK / 256; k++) {
        u64 r = base + (u64)k * 256 + threadIdx.x;
        c += seg_start(head, r, n) ? 1u : 0u;
    }
    u32 tot;
    bfq_block_exscan32(c, sh, &tot);
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}

// Writes one entry per segment of >= 2 rows: start row (40 bits) | size << 40, size = 2..64,
// or 0 when the segment is longer than a wavefront.  Sizes come from a head-bit mask of the
// chunk (+128 rows of look-ahead) kept in LDS.
__global__ __launch_bounds__(256) void k_seg_write(const u8 *__restrict__ head, u64 n, const u64 *__restrict__ blockBase,
                                                   u64 *__restrict__ seglist)
{
    __shared__ u32 sh[4];
    __shared__ u64 hb[SG_CHUNK / 64 + 2];
    const u32 lane = bfq_lane(), w = threadIdx.x >> 6;
    u64 base = (u64)blockIdx.x * SG_CHUNK;
    for (u32 g = w; g < SG_CHUNK / 64 + 2; g += 4) {
        u64 r = base + (u64)g * 64 + lane;
        bool h = (r >= n) ? true : (head[r] != 0);          // rows past the end close the last segment
        u64 m = __ballot(h);
        if (lane == 0) hb[g] = m;
    }
    __syncthreads();
    u64 out = blockBase[blockIdx.x];
    for (int k = 0; k < SG_CHUNK / 256; k++) {               // chunk order = row order
        u32 li = k * 256 + threadIdx.x;
        u64 r = base + li;
        bool s = (r + 1 < n) && ((hb[li >> 6] >> (li & 63)) & 1ull) && !((hb[(li + 1) >> 6] >> ((li + 1) & 63)) & 1ull);
        u32 size = 0;
        if (s) {
            u32 idx = li + 1;
            u64 wd = hb[idx >> 6] >> (idx & 63);
            u32 nxt;
            if (wd) nxt = idx + (u32)__builtin_ctzll(wd);
            else {
                u64 w2 = hb[(idx >> 6) + 1];
                nxt = w2 ? (((idx >> 6) + 1) << 6) + (u32)__builtin_ctzll(w2) : 0xFFFFu;
            }
            size = nxt - li;
            if (size > 64) size = 0;                         // handled by k_refine_big
        }
        u32 tot;
        u32 ex = bfq_block_exscan32(s ? 1u : 0u, sh, &tot);
        if (s) seglist[out + ex] = r | ((u64)size << 40);
        out += tot;
    }
}

__device__ __forceinline__ u32 bfq_wave_max32(u32 v)
{
    u32 lane = bfq_lane();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        u32 o = (u32)__builtin_amdgcn_ds_bpermute((int)((lane ^ d) << 2), (int)v);
        v = o > v ? o : v;
    }
    return v;
}

// ---- segments of <= 64 rows: a wavefront takes 64 list entries, packs as many segments as
// fit into its 64 lanes (one row per lane) and refines them together, round by round:
//   fetch the next 21-symbol word of every open row (prefetched one round ahead),
//   stable rank inside each sub-segment by counting over lane offsets (ds_bpermute),
//   one ds_permute of the payload, new sub-segment heads and their LCP from the words.
__global__ __launch_bounds__(256) void k_refine_wave(const u64 *__restrict__ seglist, u64 nseg, u64 *__restrict__ vals,
                                                     u16 *__restrict__ lcp, const u64 *__restrict__ text3, u64 n,
                                                     u32 depth0, u64 *__restrict__ biglist, DevCounters *cnt)
{
    const u32 lane = bfq_lane();
    const u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    const u64 nbatch = (nseg + 63) >> 6;
    for (u64 batch = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; batch < nbatch; batch += nwaves) {
        u64 idx = batch * 64 + lane;
        u64 ent = (idx < nseg) ? seglist[idx] : 0ull;
        u64 segStart = ent & BFQ_POS_MASK;
        u32 segSize = (idx < nseg) ? (u32)(ent >> 40) & 0x7Fu : 0u;
        if (idx < nseg && segSize == 0) biglist[atomicAdd(&cnt->bigCount, 1ull)] = segStart;
        u32 incl = bfq_wave_incscan32(segSize);
        u32 excl = incl - segSize;
        u32 done = 0;
        while (done < 64) {
            u32 base = (u32)__builtin_amdgcn_readlane((int)excl, (int)done);
            u64 take = __ballot(lane >= done && incl - base <= 64u);
            u32 ntake = (u32)__popcll(take);                                    // >= 1
            u32 rows = (u32)__builtin_amdgcn_readlane((int)incl, (int)(done + ntake - 1)) - base;
            if (rows == 0) { done += ntake; continue; }
            // rows -> lanes
            u64 myRow = 0;
            int sublo = 0, subhi = 0, seglo = 0;
            for (u32 t = 0; t < ntake; t++) {
                int k = (int)(done + t);
                u32 sz = (u32)__builtin_amdgcn_readlane((int)segSize, k);
                if (!sz) continue;                                               // uniform
                int hp = (int)((u32)__builtin_amdgcn_readlane((int)excl, k) - base);
                u64 st = bfq_readlane64(segStart, k);
                if ((int)lane >= hp && (int)lane < hp + (int)sz) { myRow = st + (u64)((int)lane - hp); sublo = hp; subhi = hp + (int)sz; seglo = hp; }
            }
            const bool act = lane < rows;
            u64 v = act ? vals[myRow] : 0ull;
            u32 mylcp = LCP_PENDING;                                            // positional: LCP(row-1,row)
            u64 unres = __ballot(act);
            u32 depth = depth0;
            u64 Wn = act ? bfq_key_at(text3, bfq_val_pos(v) + depth) : 0ull;
            while (unres) {
                const bool un = (unres >> lane) & 1ull;
                u64 W = Wn;
                Wn = (un && !bfq_key_has_term(W)) ? bfq_key_at(text3, bfq_val_pos(v) + depth + BFQ_SYMS_PER_WORD) : 0ull;
                u32 maxsz = bfq_wave_max32(un ? (u32)(subhi - sublo) : 0u);
                int c = 0;
                u64 Wu = W, Wd = W;                                             // W of lane+d / lane-d, shifted one lane per step
                for (u32 d = 1; d < maxsz; d++) {
                    int up = (int)lane + (int)d, dn = (int)lane - (int)d;
                    Wu = bfq_from_next_lane(Wu);
                    Wd = bfq_from_prev_lane(Wd);
                    if (un && up < subhi && Wu < W) c++;
                    if (un && dn >= sublo && Wd <= W) c++;
                }
                int np = un ? sublo + c : (int)lane;
                v = bfq_permute64(v, np);
                W = bfq_permute64(W, np);
                Wn = bfq_permute64(Wn, np);
                u64 Wprev = bfq_bpermute64(W, (int)lane - 1);
                bool newhead = un && ((int)lane == sublo || W != Wprev || bfq_key_has_term(W));
                if (un && (int)lane != sublo && newhead) mylcp = depth + (u32)bfq_key_lcp(Wprev, W);
                u64 heads = __ballot(newhead || !un);
                sublo = 63 - __clzll((long long)(heads & le));
                u64 above = heads & ~le;
                subhi = above ? __builtin_ctzll(above) : 64;
                unres = __ballot(un && (subhi - sublo > 1));
                depth += BFQ_SYMS_PER_WORD;
            }
            if (act) {
                vals[myRow] = v;
                if ((int)lane != seglo) lcp[myRow] = (u16)mylcp;
            }
            done += ntake;
        }
    }
}

// ---- larger segments: one workgroup each, bitonic network in global memory -------
// full-suffix order beyond the first 21 symbols; ties (identical suffixes) by position
__device__ bool suffix_less(const u64 *__restrict__ text3, u64 pa, u64 pb, u32 depth0)
{
    for (u32 d = depth0;; d += BFQ_SYMS_PER_WORD) {
        u64 a = bfq_key_at(text3, pa + d), b = bfq_key_at(text3, pb + d);
        if (a != b) return a < b;
        if (bfq_key_has_term(a)) return pa < pb;
    }
}
__device__ u32 suffix_lcp(const u64 *__restrict__ text3, u64 pa, u64 pb, u32 depth0)
{
    for (u32 d = depth0;; d += BFQ_SYMS_PER_WORD) {
        u64 a = bfq_key_at(text3, pa + d), b = bfq_key_at(text3, pb + d);
        if (a != b || bfq_key_has_term(a)) return d + (u32)bfq_key_lcp(a, b);
    }
}

__device__ __forceinline__ void big_step(u64 *a, u64 g, u64 j, const u64 *__restrict__ text3, u32 depth0)
{
    for (u64 i = threadIdx.x; i < g; i += 256) {
        u64 l = i ^ j;
        if (l > i && l < g) {                      // rows >= g are a virtual +inf padding
            u64 va = a[i], vb = a[l];
            if (suffix_less(text3, bfq_val_pos(vb), bfq_val_pos(va), depth0)) { a[i] = vb; a[l] = va; }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_refine_big(const u64 *__restrict__ biglist, u64 nbig, u64 *__restrict__ vals,
                                                    const u8 *__restrict__ head, u16 *__restrict__ lcp,
                                                    const u64 *__restrict__ text3, u64 n, u32 depth0)
{
    __shared__ u64 shEnd;
    for (u64 bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
    const u64 s = biglist[bi];
    __syncthreads();
    if (threadIdx.x == 0) shEnd = ~0ull;
    __syncthreads();
    for (u64 base = s + 1;; base += 256) {         // first head after s = end of the segment
        u64 i = base + threadIdx.x;
        bool h = (i >= n) ? true : (head[i] != 0);
        if (h) atomicMin(&shEnd, i);
        __syncthreads();
        bool done = (shEnd != ~0ull);
        __syncthreads();
        if (done) break;                           // uniform
    }
    const u64 g = shEnd - s;
    u64 *a = vals + s;
    u64 P = 1;
    while (P < g) P <<= 1;
    // bitonic network with ascending comparators only: flip (i <-> i^(k-1)) then
    // disperse (i <-> i^j, j = k/4 .. 1); the +inf padding beyond g never moves
    for (u64 k = 2; k <= P; k <<= 1) {
        big_step(a, g, k - 1, text3, depth0);
        for (u64 j = k >> 2; j >= 1; j >>= 1) big_step(a, g, j, text3, depth0);
    }
    for (u64 i = 1 + threadIdx.x; i < g; i += 256)
        lcp[s + i] = (u16)suffix_lcp(text3, bfq_val_pos(a[i - 1]), bfq_val_pos(a[i]), depth0);
    }
}

// eBWT byte and permuted quality of every row, from the sort payload
__global__ __launch_bounds__(256) void k_emit_bwt(const u64 *__restrict__ vals, u64 n, u32 termOut, u8 *__restrict__ bwt,
                                                  u8 *__restrict__ qs)
{
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x) {
        u64 v = vals[r];
        u32 code = bfq_val_code(v);
        bwt[r] = code ? bfq_code_sym(code) : (u8)termOut;
        qs[r] = (u8)bfq_val_qual(v);
    }
}

void bfq_refine(bfq_ctx *c, const u64 *keys, u64 *vals, const u64 *text3, u64 n, int keySyms, u16 *lcp, bfq_stats *st)
{
    const int lowbit = 3 * (BFQ_SYMS_PER_WORD - keySyms);
    if (!n) return;
    size_t m = c->mark();
    u8 *head = c->alloc<u8>(n + 64);
    KLAUNCH(c, K_SEG_FLAGS, 19.0 * (double)n, k_seg_flags, bfq_grid(n, 256), 256, keys, n, lowbit, head, lcp);
    u64 nchunks = ceil_div(n, SG_CHUNK);
    u32 *counts = c->alloc<u32>(nchunks);
    u64 *bases = c->alloc<u64>(nchunks);
    u64 *d_total = c->alloc<u64>(1);
    KLAUNCH(c, K_SEG_COMPACT, (double)n, k_seg_count, nchunks, 256, (const u8 *)head, n, counts);
    bfq_exscan_u32(c, counts, bases, nchunks, d_total);
    u64 nseg = 0;
    HIP_CHECK(hipMemcpyAsync(&nseg, d_total, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (st) st->n_segments = nseg;
    if (nseg) {
        u64 *seglist = c->alloc<u64>(nseg);
        u64 *biglist = c->alloc<u64>(nseg);
        KLAUNCH(c, K_SEG_COMPACT, (double)n + 8.0 * (double)nseg, k_seg_write, nchunks, 256, (const u8 *)head, n,
                (const u64 *)bases, seglist);
        KLAUNCH(c, K_REFINE_WAVE, 26.0 * (double)n, k_refine_wave, bfq_grid(ceil_div(nseg, 64), 4), 256,
                (const u64 *)seglist, nseg, vals, lcp, text3, n, (u32)keySyms, biglist, c->d_cnt);
        u64 nbig = 0;
        HIP_CHECK(hipMemcpyAsync(&nbig, &c->d_cnt->bigCount, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        c->sync();
        if (st) st->n_big_segments = nbig;
        if (nbig)
            KLAUNCH(c, K_REFINE_BIG, 0.0, k_refine_big, bfq_grid(nbig, 1), 256, (const u64 *)biglist, nbig, vals, (const u8 *)head,
                    lcp, text3, n, (u32)keySyms);
    }
    c->release(m);
}

void bfq_emit_bwt(bfq_ctx *c, const u64 *vals, u64 n, int termOut, u8 *bwt, u8 *qs)
{
    if (!n) return;
    KLAUNCH(c, K_EMIT, 10.0 * (double)n, k_emit_bwt, bfq_grid(n, 256), 256, vals, n, (u32)(termOut & 0xFF), bwt, qs);
}
