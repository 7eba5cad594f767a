// k_bfs.hip -- the LCP array deduced from the eBWT alone: what bfq_int does when it is given only
// OUT.bwt / OUT.bwt.qs (detect_minima, bfq_int.cpp:183-300, navigating the suffix tree through
// dna_bwt_n::LF(range) / next_leaves / next_nodes, dna_bwt_n.hpp:119-148,312-362).
//
// MI355X formulation: level-synchronous interval refinement (Beller, Gog, Ohlebusch, Schnattinger:
// "Computing the longest common prefix array based on the Burrows-Wheeler transform").  An interval
// [lb, rb] of the rows whose suffixes start with a string w is extended to the left by every symbol c
// (two rank queries per end, all five symbols from ONE 64-byte rank block each): the rows starting with
// cw are [lb', rb'], and LCP[rb'+1] -- the boundary after them -- equals |w| unless a shorter string
// already claimed it.  An interval is enqueued only when it claims a boundary, so every LCP entry is
// written exactly once and the total work is n intervals, processed level by level (level = |w|),
// each level one flat kernel over the queue segment the previous level appended.
// The queue is ONE array of n entries used as a log (n enqueues in total, none overwritten).
//
// Two families of intervals are refined side by side:
//   * base intervals: rows starting with a string w of bases.  A child cw is enqueued only when it claims the
//     boundary LCP[rb'+1] (the classic pruning).
//   * leaf blocks: the rows "u#" of the identical suffixes u followed by a terminator (the reference's sa_leaf,
//     bfq_int.cpp:139-145, dna_bwt_n.hpp:312-335).  Two suffixes never match on their terminators --
//     LCP(u#_i, u#_j) = |u|, the convention of bfq_int.cpp:139-145 and bfq_ext.cpp:377-392 -- so a child block
//     "cu#" = [lb', rb'] has LCP[lb'+1 .. rb'] = |cu| inside and claims LCP[rb'+1] = |cu| when nobody did;
//     it is enqueued when it claimed or holds at least two rows.  The block of all N terminator suffixes is the
//     leaf of level 1.  Blocks are handled as ranges, so NO assumption is made on the order of identical
//     suffixes (or of the terminator rows) in the given eBWT -- any tool's tie order is accepted, like the
//     reference, which sees one terminator symbol.
// Rows whose eBWT symbol is the terminator are never extended (nothing precedes a whole read).
// tests/bfs_model.py is the executable model of this algorithm (checked against LCPs computed from the decoded
// suffixes, incl. eBWTs whose ties are in inconsistent order).
//
//   rank block (64 B, rows [64 b, 64 b + 64)): u64 count of A, C, G, N, T before the block, then the
//   three bit planes of the rows' symbol codes (# 0, A 1, C 2, G 3, N 4, T 5).
//   queue entry: lb (38 bits) | (rb - lb) (25 bits) | leaf (1 bit); longer intervals go to a small side list.
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_rankblk.h"

#define BQ_LB_BITS 38
#define BQ_LB_MASK ((1ull << BQ_LB_BITS) - 1ull)
#define BQ_LEN_BITS 25
#define BQ_LEN_MAX ((1ull << BQ_LEN_BITS) - 1ull)            // rb - lb values up to here fit the packed entry
#define BQ_LEAF (1ull << 63)
#define BQ_FILL_INLINE 64                                    // leaf blocks up to this many inner boundaries are filled by their wavefront
#define LCP_UNSET 0xFFFFu

struct BfsArgs {
    const u64 *rank;        // [n/64 + 2][8]
    u64 F[6];               // first row of every symbol's suffixes (# A C G N T)
    u16 *lcp;               // [n + 1]
    u64 *queue;             // [n + 64]; ring mode (qsize != 0): [qsize], positions taken modulo its size
    u64 qsize;
    u64 qlimit;             // log mode: positions below this one have an entry to be stored to
    u64 *tail;              // [0] next free queue slot, [1] / [2] lengths of the two side lists, [3] LCP entries written, [4] fill list length
    u64 *side[2];           // long intervals (lb, rb | leaf) of the current / next level
    u64 sideCap;
    u64 *fill;              // long leaf blocks (lb, rb) whose inner boundaries k_bfs_fill writes
    u64 fillCap;
    u64 n;
};

// rank blocks from the eBWT bytes: one wavefront per group of 256 rows (4 blocks); `scanned` = occurrences of every
// symbol before the group (k_lf_count + scans, k_rank.hip)
__global__ __launch_bounds__(256) void k_rankblocks(const u8 *__restrict__ bwt, u64 n, u32 term, const u64 *__restrict__ scanned,
                                                    u64 ngroups, u64 *__restrict__ rank, DevCounters *cnt)
{
    const u32 lane = bfq_lane();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    bool bad = false;
    for (u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; g < ngroups; g += nwaves) {
        u64 base = (lane >= 1 && lane <= 5) ? scanned[(u64)lane * ngroups + g] : 0ull;   // lane c: count of code c so far
#pragma unroll 1
        for (u32 sb = 0; sb < 4; sb++) {
            const u64 r = g * 256 + sb * 64 + lane;
            u32 code = 0;
            if (r < n) {
                const u32 ch = bwt[r];
                code = (ch == term) ? 0u : bfq_base_code((u8)ch);
                if (code == BFQ_CODE_INVALID) { bad = true; code = 4; }
            }
            const u64 p0 = __ballot(code & 1u), p1 = __ballot(code & 2u), p2 = __ballot(code & 4u);
            // mask of the rows holding code `lane` (lanes 1..5), from the planes
            const u64 m = ((lane & 1u) ? p0 : ~p0) & ((lane & 2u) ? p1 : ~p1) & ((lane & 4u) ? p2 : ~p2);
            u64 *blk = rank + ((g * 4 + sb) << 3);
            if (lane >= 1 && lane <= 5) blk[lane - 1] = base;
            if (lane == 5) blk[5] = p0;
            if (lane == 6) blk[6] = p1;
            if (lane == 7) blk[7] = p2;
            base += (u64)__popcll(m);
        }
    }
    if (bad) atomicAdd(&cnt->errSymbol, 1ull);
}

// Extends the intervals of one level.  SIDE: the (few) long intervals of the side list instead of the queue segment.
#define BQ_STAGE 4096                                        // children staged in LDS per queue reservation
template <bool SIDE>
__global__ __launch_bounds__(256) void k_bfs_level(BfsArgs a, u64 qbeg, u64 qend, u32 level, int cur)
{
    // The queue tail is ONE address for the whole device: a reservation per wavefront (n / 64 atomics) serialises
    // there and was 80 % of the kernel.  A workgroup stages its children in LDS over several rounds and reserves
    // once per ~4000 of them.
    __shared__ u64 stage[BQ_STAGE];
    __shared__ u64 sBase;
    __shared__ u32 scan[4];
    u32 used = 0;                                                  // uniform: every thread sees the same totals
    const u32 lane = bfq_lane();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 cnt = SIDE ? a.tail[1 + cur] : qend - qbeg;
    const u64 rounds = (cnt + stride - 1) / stride;
    u64 written = 0;                                               // LCP entries this thread wrote or handed to the fill list
    for (u64 it = 0; it < rounds; it++) {                          // whole waves stay together: wave-wide enqueue below
        const u64 i = it * stride + (u64)blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = i < cnt;
        u64 lb = 0, rb = 0;
        bool leaf = false;
        if (valid) {
            if (SIDE) { lb = a.side[cur][2 * i]; rb = a.side[cur][2 * i + 1]; leaf = (rb & BQ_LEAF) != 0; rb &= ~BQ_LEAF; }
            else { const u64 e = __builtin_nontemporal_load(a.queue + (a.qsize ? (qbeg + i) % a.qsize : qbeg + i)); lb = e & BQ_LB_MASK; rb = lb + ((e >> BQ_LB_BITS) & BQ_LEN_MAX); leaf = (e & BQ_LEAF) != 0; }
        }
        u32 nkids = 0;
        u64 kid[5];                                                // packed children to enqueue
        u64 flb = 0, frb = 0;                                      // one pending inline fill per thread: (flb, frb]
        if (valid) {
            u64 ol[5], orr[5];
            {
                const u64 bl = lb >> 6, br = (rb + 1) >> 6;
                const RankBlk B0 = load_blk(a.rank, bl);
                occ5(B0, lb, ol);
                if (br == bl) occ5(B0, rb + 1, orr);               // short intervals: both ends in one block
                else { const RankBlk B1 = load_blk(a.rank, br); occ5(B1, rb + 1, orr); }
            }
            // all boundary probes first (independent loads), then the claims
            u64 nlb[5], nrb[5];
            u32 probe[5];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                nlb[c] = a.F[c + 1] + ol[c]; nrb[c] = a.F[c + 1] + orr[c] - 1;
                probe[c] = (orr[c] > ol[c]) ? (u32)__builtin_nontemporal_load(a.lcp + nrb[c] + 1) : 0u;
            }
#pragma unroll
            for (int c = 0; c < 5; c++) {
                if (orr[c] > ol[c]) {
                    const bool claim = probe[c] == LCP_UNSET;
                    if (claim) { a.lcp[nrb[c] + 1] = (u16)level; written++; }
                    const bool wide = leaf && nrb[c] > nlb[c];
                    if (wide) {                                    // inner boundaries of the block of identical suffixes
                        const u64 inner = nrb[c] - nlb[c];
                        written += inner;
                        if (inner <= 2) { a.lcp[nlb[c] + 1] = (u16)level; a.lcp[nrb[c]] = (u16)level; }
                        else if (inner <= BQ_FILL_INLINE && frb == flb) { flb = nlb[c]; frb = nrb[c]; }
                        else if (inner <= BQ_FILL_INLINE) { for (u64 p = nlb[c] + 1; p <= nrb[c]; p++) a.lcp[p] = (u16)level; }   // a second block of the same parent: rare
                        else {
                            const u64 k = atomicAdd((unsigned long long *)&a.tail[4], 1ull);
                            if (k < a.fillCap) { a.fill[2 * k] = nlb[c]; a.fill[2 * k + 1] = nrb[c]; }
                        }
                    }
                    if (claim || wide) {
                        if (nrb[c] - nlb[c] <= BQ_LEN_MAX) kid[nkids++] = nlb[c] | ((nrb[c] - nlb[c]) << BQ_LB_BITS) | (leaf ? BQ_LEAF : 0ull);
                        else {
                            const u64 k = atomicAdd((unsigned long long *)&a.tail[1 + (cur ^ 1)], 1ull);
                            if (k < a.sideCap) { a.side[cur ^ 1][2 * k] = nlb[c]; a.side[cur ^ 1][2 * k + 1] = nrb[c] | (leaf ? BQ_LEAF : 0ull); }
                        }
                    }
                }
            }
        }
        // inline fills, one block at a time by the whole wavefront
        for (u64 pend = __ballot(frb > flb); pend; pend &= pend - 1) {
            const int src = __builtin_ctzll(pend);
            const u64 l0 = bfq_readlane64(flb, src), r0 = bfq_readlane64(frb, src);
            if (l0 + 1 + lane <= r0) a.lcp[l0 + 1 + lane] = (u16)level;
        }
        // children -> LDS stage; flushed with one reservation when the next round might not fit
        u32 tot;
        const u32 ex = bfq_block_exscan32(nkids, scan, &tot);
        if (used + tot > BQ_STAGE) {
            if (threadIdx.x == 0) sBase = atomicAdd((unsigned long long *)&a.tail[0], (unsigned long long)used);
            __syncthreads();
            // (a byte sequence that is no eBWT can enqueue more than the n intervals a real one has: the stores stop at the
            // queue's end, the host sees the tail beyond it and reports BFQ_E_NOT_EBWT)
            for (u32 j = threadIdx.x; j < used; j += 256) {
                // ring: never onto an unread entry (they start at qbeg; the host launches only as many parents as have room for
                // five children each, so the flag is an assertion)
                if (a.qsize) { if (sBase + j - qbeg < a.qsize) __builtin_nontemporal_store(stage[j], a.queue + (sBase + j) % a.qsize); else a.tail[5] = 1; }
                else if (sBase + j < a.qlimit) __builtin_nontemporal_store(stage[j], a.queue + sBase + j);
            }
            __syncthreads();
            used = 0;
        }
        for (u32 k = 0; k < nkids; k++) stage[used + ex + k] = kid[k];
        used += tot;
    }
    __syncthreads();
    if (used) {
        if (threadIdx.x == 0) sBase = atomicAdd((unsigned long long *)&a.tail[0], (unsigned long long)used);
        __syncthreads();
        for (u32 j = threadIdx.x; j < used; j += 256) {
            if (a.qsize) { if (sBase + j - qbeg < a.qsize) __builtin_nontemporal_store(stage[j], a.queue + (sBase + j) % a.qsize); else a.tail[5] = 1; }
            else if (sBase + j < a.qlimit) __builtin_nontemporal_store(stage[j], a.queue + sBase + j);
        }
    }
    const u64 wsum = bfq_readlane64(bfq_wave_incscan64(written), 63);
    if (lane == 0 && wsum) atomicAdd((unsigned long long *)&a.tail[3], (unsigned long long)wsum);
}

// inner boundaries of the long leaf blocks listed by the level kernel
__global__ __launch_bounds__(256) void k_bfs_fill(BfsArgs a, u64 count, u32 level)
{
    for (u64 bi = blockIdx.x; bi < count; bi += gridDim.x) {
        const u64 lb = a.fill[2 * bi], rb = a.fill[2 * bi + 1];
        for (u64 p = lb + 1 + threadIdx.x; p <= rb; p += blockDim.x) a.lcp[p] = (u16)level;
    }
}

// level 0: the root interval's children -- the block of the N terminator suffixes (a leaf), one interval per base
__global__ __launch_bounds__(256) void k_bfs_init(BfsArgs a, u64 N)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i <= N; i += stride)
        a.lcp[i] = 0;                                              // row 0, the boundaries between terminator suffixes, the first base suffix
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u64 q = 0, ns = 0, wr = 0;
        if (N) {
            wr += N;
            if (N - 1 <= BQ_LEN_MAX) a.queue[q++] = 0ull | ((N - 1) << BQ_LB_BITS) | BQ_LEAF;
            else { a.side[1][2 * ns] = 0; a.side[1][2 * ns + 1] = (N - 1) | BQ_LEAF; ns++; }
        }
        for (int c = 1; c <= 5; c++) {
            const u64 lb = a.F[c], end = (c == 5) ? a.n : a.F[c + 1];
            if (end > lb) {
                a.lcp[end] = 0; wr++;
                if (end - 1 - lb <= BQ_LEN_MAX) a.queue[q++] = lb | ((end - 1 - lb) << BQ_LB_BITS);
                else { a.side[1][2 * ns] = lb; a.side[1][2 * ns + 1] = end - 1; ns++; }
            }
        }
        a.lcp[a.n] = 0;
        a.tail[0] = q; a.tail[1] = 0; a.tail[2] = ns; a.tail[3] = wr; a.tail[4] = 0; a.tail[5] = 0;   // level 1 reads side list 1
    }
}

u64 *bfq_symbol_scans(bfq_ctx *c, const u8 *bwt, u64 n, int term, const u32 *gcntIn, u32 *gcntOut);   // k_rank.hip

// rank blocks of an eBWT (rank: (n / 256 + 1) * 4 + 2 blocks of 8 words); the symbol totals land in d_cnt->tot
void bfq_rank_blocks(bfq_ctx *c, const u8 *bwt, u64 n, int term, const u64 *scanned, u64 *rank)
{
    const u64 ngroups = n / 256 + 1;
    KLAUNCH(c, K_RANK_BUILD, 2.0 * (double)n, k_rankblocks, bfq_grid(ngroups, 4), 256, bwt, n, (u32)(term & 0xFF), scanned, ngroups, rank, c->d_cnt);
}

// lcp[0..n): LCP array of the eBWT `bwt` (device, n rows, N of them terminators).  Workspace: n bytes of rank blocks,
// 8 n bytes of queue (released on return).  rankGiven: the rank blocks exist already (and the symbol totals are in
// c->h_cnt).  ringEntries != 0 (k_compact.hip, under a cap): the queue is a ring of that many entries instead of a log of n.
// Only the unread part of the level being read and the level being written are live.  A level is then launched in chunks:
// as many parents as the free entries hold five children for (an interval has at most five); when they are done their
// slots are free.  Two consecutive levels that do not fit the ring at all (little coverage: most LCP values lie within a few
// levels of log4 n) move the queue to pinned host memory -- the kernels read and write it there, 16 bytes per interval
// over the link -- rather than fail.
void bfq_lcp_from_bwt(bfq_ctx *c, const u8 *bwt, u64 n, u64 N, int term, u16 *lcp, u32 *gcntOut, const u64 *rankGiven, u64 ringEntries)
{
    if (!n) return;
    size_t mk = c->mark();
    const u64 ngroups = n / 256 + 1;
    BfsArgs a;
    const u64 *rank = rankGiven;
    if (!rankGiven) {
        u64 *scanned = bfq_symbol_scans(c, bwt, n, term, nullptr, gcntOut);   // the counts stay for the LF table build
        u64 *rk = c->alloc<u64>((ngroups * 4 + 2) * 8);
        KLAUNCH(c, K_RANK_BUILD, 2.0 * (double)n, k_rankblocks, bfq_grid(ngroups, 4), 256, bwt, n, (u32)(term & 0xFF), (const u64 *)scanned, ngroups, rk, c->d_cnt);
        c->fetchCounters();                                        // symbol totals -> F
        rank = rk;
    }
    {
        u64 acc = 0;
        for (int s = 0; s < 6; s++) { a.F[s] = acc; acc += c->h_cnt.tot[s]; }
        if (acc != n || c->h_cnt.tot[0] != N) throw BfqError{BFQ_E_NOT_EBWT, "symbol counts do not add up to the eBWT"};
    }
    if (ringEntries && ringEntries < 64) ringEntries = 64;
    if (ringEntries >= n + 64) ringEntries = 0;                    // as large as the log: be the log
    a.rank = rank; a.lcp = lcp; a.n = n;
    a.qsize = ringEntries; a.qlimit = n + 64;
    a.queue = c->alloc<u64>(ringEntries ? ringEntries : n + 64);
    a.tail = c->alloc<u64>(8);
    a.sideCap = (n >> BQ_LEN_BITS) + 16;
    a.side[0] = c->alloc<u64>(2 * a.sideCap); a.side[1] = c->alloc<u64>(2 * a.sideCap);
    a.fillCap = n / (BQ_FILL_INLINE + 1) + 16;                     // listed blocks are disjoint and hold more than BQ_FILL_INLINE rows each
    a.fill = c->alloc<u64>(2 * a.fillCap);
    struct HostQueue { u64 *p = nullptr; ~HostQueue() { if (p) (void)hipHostFree(p); } } hostQ;
    HIP_CHECK(hipMemsetAsync(lcp, 0xFF, 2 * (n + 1), c->stream));
    KLAUNCH(c, K_BFS, 2.0 * (double)N, k_bfs_init, bfq_grid(N + 1, 256), 256, a, N);
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    auto fetch = [&] {
        HIP_CHECK(hipMemcpyAsync(t, a.tail, 48, hipMemcpyDeviceToHost, c->stream));
        c->sync();
    };
    fetch();
    u64 qbeg = 0, qend = t[0], nside = t[2];
    int cur = 1;
    u32 chunksMax = 1;
    // ring -> log in pinned host memory; [from, t[0]) are the live entries
    auto to_host = [&](u64 from, u32 level) {
        const u64 live = t[0] - from, room = live + (n - (t[3] < n ? t[3] : n)) + 1024;   // every further entry writes an LCP value
        void *hp = nullptr;
        if (hipHostMalloc(&hp, room * 8, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            throw BfqError{BFQ_E_NOMEM, "interval refinement: two levels do not fit the queue the workspace cap leaves room for, nor pinned host memory; raise bfq_params.ws_cap_mib / BFQ_WS_CAP"};
        }
        hostQ.p = (u64 *)hp;
        for (u64 o = from; o < t[0];) {
            const u64 ri = o % a.qsize, len = (t[0] - o < a.qsize - ri) ? t[0] - o : a.qsize - ri;
            HIP_CHECK(hipMemcpyAsync(hostQ.p + (o - from), a.queue + ri, len * 8, hipMemcpyDeviceToHost, c->stream));
            o += len;
        }
        c->sync();
        void *dp = nullptr;
        HIP_CHECK(hipHostGetDevicePointer(&dp, hp, 0));
        if (bfq_env().trace)
            fprintf(stderr, "[bfq] interval refinement: level %u holds %llu intervals, the ring %llu: queue moved to %.2f GB of pinned host memory\n",
                    level, (unsigned long long)live, (unsigned long long)a.qsize, room * 8 / 1e9);
        a.queue = (u64 *)((uintptr_t)dp - (uintptr_t)from * 8);    // indexed by absolute position from here on
        a.qsize = 0; a.qlimit = from + room;
    };
    auto check = [&] {
        if (t[0] > n + 32 || t[0] + 32 > a.qlimit || t[1 + (cur ^ 1)] > a.sideCap || t[4] > a.fillCap)
            throw BfqError{BFQ_E_NOT_EBWT, "interval refinement overran the eBWT: not a BWT"};
        if (t[5]) throw BfqError{BFQ_E_NOMEM, "interval refinement: ring queue overrun"};
    };
    for (u32 level = 1; qend > qbeg || nside; level++) {
        if (level > BFQ_MAX_READ_LEN + 2) throw BfqError{BFQ_E_TOO_LONG, "LCP beyond BFQ_MAX_READ_LEN (or not an eBWT)"};
        HIP_CHECK(hipMemsetAsync(a.tail + 1 + (cur ^ 1), 0, 8, c->stream));
        // per interval: two 64-B rank blocks, per child one LCP probe + store, 8 B of queue in and out
        u64 pb = qbeg;
        if (a.qsize) {
            const u64 sideRoom = 5 * nside;                        // the long intervals of the side list are extended last
            if (a.qsize - (t[0] - pb) < sideRoom) to_host(pb, level);
            u32 chunks = 0;
            while (a.qsize && pb < qend) {
                const u64 rem = qend - pb, room = a.qsize - (t[0] - pb) - sideRoom;
                u64 chunk = room / 5;
                if (chunk < rem && chunk < (rem / 64 > 4096 ? rem / 64 : 4096)) { to_host(pb, level); break; }   // (not in a crawl)
                if (chunk > rem) chunk = rem;
                KLAUNCH(c, K_BFS, 160.0 * (double)chunk, k_bfs_level<false>, bfq_grid(chunk, 256 * 8), 256, a, pb, pb + chunk, level, cur);
                pb += chunk;
                if (pb < qend) { fetch(); check(); }
                chunks++;
            }
            if (chunks > chunksMax) chunksMax = chunks;
        }
        if (qend > pb)
            KLAUNCH(c, K_BFS, 160.0 * (double)(qend - pb), k_bfs_level<false>, bfq_grid(qend - pb, 256 * 8), 256, a, pb, qend, level, cur);
        if (nside) KLAUNCH(c, K_BFS, 0.0, k_bfs_level<true>, bfq_grid(nside, 256), 256, a, qend, qend, level, cur);
        fetch();
        check();
        if (t[4]) {
            KLAUNCH(c, K_BFS, 0.0, k_bfs_fill, (unsigned)(t[4] < 4096 ? t[4] : 4096), 256, a, t[4], level);
            HIP_CHECK(hipMemsetAsync(a.tail + 4, 0, 8, c->stream));
        }
        qbeg = qend; qend = t[0];
        nside = t[1 + (cur ^ 1)];
        cur ^= 1;
    }
    if (bfq_env().trace && ringEntries)
        fprintf(stderr, "[bfq] interval refinement: ring of %llu entries for %llu rows, at most %u launches per level%s\n",
                (unsigned long long)ringEntries, (unsigned long long)n, chunksMax, hostQ.p ? ", finished in host memory" : "");
    // every LCP entry 1..n is written exactly once
    if (t[3] != n) throw BfqError{BFQ_E_NOT_EBWT, "interval refinement does not cover the eBWT: not a BWT of a read collection"};
    c->sync();
    c->release(mk);
}
