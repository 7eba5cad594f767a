// k_refine.hip -- finishes the suffix order after the 21-symbol radix sort and
// produces the LCP array, then the eBWT / permuted-quality bytes.
//
// After the sort, equal keys form segments.  A key that holds a terminator is a
// complete suffix: equal such keys are identical suffixes, already in read order
// (stable sort, #_i < #_j for i < j), so they are final.  All other segments of
// >= 2 rows are refined on the following 21-symbol words of the text:
//   k_refine_wave : one wavefront per segment of <= 64 rows, all rounds in
//                   registers (rank-by-counting + ds_permute), LCP as a by-product
//   k_refine_big  : one workgroup per larger segment, bitonic network over the
//                   rows in global memory with a full suffix comparator
// LCP convention: common prefix counted on bases only, terminators never match
// (what bfq_int deduces from the BWT, bfq_int.cpp:139-181,183-300, and what
// eGap --lcp hands to bfq_ext, bfq_ext.cpp:350-412).
#include "bfq_internal.h"
#include "bfq_device.h"

#define LCP_PENDING 0xFFFFu

// head[r] = 1 if row r starts a segment; LCP of head rows comes from the keys
__global__ __launch_bounds__(256) void k_seg_flags(const u64 *__restrict__ keys, u64 n, u8 *__restrict__ head,
                                                   u16 *__restrict__ lcp)
{
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x) {
        u64 k = keys[r];
        if (r == 0) { head[0] = 1; lcp[0] = 0; continue; }
        u64 kp = keys[r - 1];
        bool h = (k != kp) || bfq_key_has_term(k);
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

__global__ __launch_bounds__(256) void k_seg_write(const u8 *__restrict__ head, u64 n, const u64 *__restrict__ blockBase,
                                                   u64 *__restrict__ seglist)
{
    __shared__ u32 sh[4];
    u64 base = (u64)blockIdx.x * SG_CHUNK;
    u64 out = blockBase[blockIdx.x];
    for (int k = 0; k < SG_CHUNK / 256; k++) {           // chunk order = row order
        u64 r = base + (u64)k * 256 + threadIdx.x;
        bool s = seg_start(head, r, n);
        u32 tot;
        u32 ex = bfq_block_exscan32(s ? 1u : 0u, sh, &tot);
        if (s) seglist[out + ex] = r;
        out += tot;
    }
}

// ---- one wavefront per segment of <= 64 rows --------------------------------------
__global__ __launch_bounds__(256) void k_refine_wave(const u64 *__restrict__ seglist, u64 nseg, u64 *__restrict__ vals,
                                                     const u8 *__restrict__ head, u16 *__restrict__ lcp,
                                                     const u64 *__restrict__ text3, u64 n, u64 *__restrict__ biglist,
                                                     DevCounters *cnt)
{
    const u32 lane = bfq_lane();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 wid = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wid < nseg; wid += nwaves) {   // wave-uniform
    u64 s = seglist[wid];
    u64 idx = s + 1 + lane;
    bool h = (idx >= n) ? true : (head[idx] != 0);
    u64 hm = __ballot(h);
    if (hm == 0) {                                            // more than 64 rows
        if (lane == 0) biglist[atomicAdd(&cnt->bigCount, 1ull)] = s;
        continue;
    }
    const int g = __builtin_ctzll(hm) + 1;                    // 2..64 rows
    const bool act = (int)lane < g;
    u64 v = act ? vals[s + lane] : 0ull;
    u64 p = bfq_val_pos(v);
    int sublo = 0;                                            // first lane of my sub-segment
    u32 mylcp = LCP_PENDING;                                  // positional: LCP(row s+lane-1, row s+lane)
    u64 unres = (g == 64) ? ~0ull : ((1ull << g) - 1ull);     // lanes whose order is still open
    u32 depth = BFQ_SYMS_PER_WORD;
    while (unres) {
        const bool un = (unres >> lane) & 1ull;
        u64 W = un ? bfq_key_at(text3, p + depth) : 0ull;
        // rank inside the sub-segment by (W, position)
        int c = 0;
        for (int y = 0; y < g; y++) {
            if (!((unres >> y) & 1ull)) continue;             // uniform
            u64 Wy = bfq_readlane64(W, y);
            u64 py = bfq_readlane64(p, y);
            int sy = __builtin_amdgcn_readlane(sublo, y);
            if (un && sy == sublo && (Wy < W || (Wy == W && py < p))) c++;
        }
        int np = un ? sublo + c : (int)lane;
        v = bfq_permute64(v, np);
        p = bfq_permute64(p, np);
        W = bfq_permute64(W, np);
        u64 Wprev = bfq_bpermute64(W, (int)lane - 1);
        bool newhead = un && ((int)lane == sublo || W != Wprev || bfq_key_has_term(W));
        if (un && (int)lane != sublo && newhead) mylcp = depth + (u32)bfq_key_lcp(Wprev, W);
        u64 heads = __ballot(newhead || !un);
        u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
        sublo = 63 - __clzll((long long)(heads & le));
        u64 above = heads & ~le;
        int subhi = above ? __builtin_ctzll(above) : 64;
        unres = __ballot(un && (subhi - sublo > 1));
        depth += BFQ_SYMS_PER_WORD;
    }
    if (act) {
        vals[s + lane] = v;
        if (lane > 0) lcp[s + lane] = (u16)mylcp;
    }
    }
}

// ---- larger segments: one workgroup each, bitonic network in global memory -------
// full-suffix order beyond the first 21 symbols; ties (identical suffixes) by position
__device__ bool suffix_less(const u64 *__restrict__ text3, u64 pa, u64 pb)
{
    for (u32 d = BFQ_SYMS_PER_WORD;; d += BFQ_SYMS_PER_WORD) {
        u64 a = bfq_key_at(text3, pa + d), b = bfq_key_at(text3, pb + d);
        if (a != b) return a < b;
        if (bfq_key_has_term(a)) return pa < pb;
    }
}
__device__ u32 suffix_lcp(const u64 *__restrict__ text3, u64 pa, u64 pb)
{
    for (u32 d = BFQ_SYMS_PER_WORD;; d += BFQ_SYMS_PER_WORD) {
        u64 a = bfq_key_at(text3, pa + d), b = bfq_key_at(text3, pb + d);
        if (a != b || bfq_key_has_term(a)) return d + (u32)bfq_key_lcp(a, b);
    }
}

__device__ __forceinline__ void big_step(u64 *a, u64 g, u64 j, const u64 *__restrict__ text3)
{
    for (u64 i = threadIdx.x; i < g; i += 256) {
        u64 l = i ^ j;
        if (l > i && l < g) {                      // rows >= g are a virtual +inf padding
            u64 va = a[i], vb = a[l];
            if (suffix_less(text3, bfq_val_pos(vb), bfq_val_pos(va))) { a[i] = vb; a[l] = va; }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_refine_big(const u64 *__restrict__ biglist, u64 nbig, u64 *__restrict__ vals,
                                                    const u8 *__restrict__ head, u16 *__restrict__ lcp,
                                                    const u64 *__restrict__ text3, u64 n)
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
        big_step(a, g, k - 1, text3);
        for (u64 j = k >> 2; j >= 1; j >>= 1) big_step(a, g, j, text3);
    }
    for (u64 i = 1 + threadIdx.x; i < g; i += 256)
        lcp[s + i] = (u16)suffix_lcp(text3, bfq_val_pos(a[i - 1]), bfq_val_pos(a[i]));
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

void bfq_refine(bfq_ctx *c, const u64 *keys, u64 *vals, const u64 *text3, u64 n, u16 *lcp, bfq_stats *st)
{
    if (!n) return;
    size_t m = c->mark();
    u8 *head = c->alloc<u8>(n + 64);
    KLAUNCH(c, K_SEG_FLAGS, 19.0 * (double)n, k_seg_flags, bfq_grid(n, 256), 256, keys, n, head, lcp);
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
        KLAUNCH(c, K_REFINE_WAVE, 26.0 * (double)n, k_refine_wave, bfq_grid(nseg, 4), 256, (const u64 *)seglist, nseg,
                vals, (const u8 *)head, lcp, text3, n, biglist, c->d_cnt);
        u64 nbig = 0;
        HIP_CHECK(hipMemcpyAsync(&nbig, &c->d_cnt->bigCount, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        c->sync();
        if (st) st->n_big_segments = nbig;
        if (nbig)
            KLAUNCH(c, K_REFINE_BIG, 0.0, k_refine_big, bfq_grid(nbig, 1), 256, (const u64 *)biglist, nbig, vals, (const u8 *)head,
                    lcp, text3, n);
    }
    c->release(m);
}

void bfq_emit_bwt(bfq_ctx *c, const u64 *vals, u64 n, int termOut, u8 *bwt, u8 *qs)
{
    if (!n) return;
    KLAUNCH(c, K_EMIT, 10.0 * (double)n, k_emit_bwt, bfq_grid(n, 256), 256, vals, n, (u32)(termOut & 0xFF), bwt, qs);
}
