// k_radix.hip -- stable LSD radix sort of the suffix records on their 40-bit key (the first 16
// symbols, two per 5-bit group: bfq_common.h), 8-bit digits, 5 passes.
//
// This is the sort at the heart of the eBWT construction that replaces
// `gsufsort --bwt --qs` (call site BFQzip.py:184): one record per read suffix.  A record is
// 12 bytes in three 32-bit arrays (bfq_common.h): the histogram pass reads only the word that
// holds the pass's digit (4 B/row), the scatter moves 12 B/row each way.  Per pass:
//   k_radix_hist    : per-workgroup digit counts (per-wave LDS histograms)
//   exclusive scan  : digit-major table -> global offsets (k_scan.hip)
//   k_radix_scatter : wave-level match ranking (ballots), per-wave LDS digit counters +
//                     prefix across waves, then each of the three words staged through LDS
//                     in tile-sorted order and written out in coalesced runs.
// HBM-bound integer work: per pass 4 B/row (hist) + 12 B read + 12 B written.
#include "bfq_internal.h"
#include "bfq_device.h"

#ifndef RS_THREADS
#define RS_THREADS 256
#endif
#define RS_WAVES (RS_THREADS / 64)
#define RS_OCC (1024 / RS_THREADS)                  // workgroups per CU (16 waves); 512-thread tiles measured 40 % slower
#define RS_ROUNDS 12                                // items per thread
#define RS_TILE (RS_THREADS * RS_ROUNDS)            // 3072 records per tile
static_assert(RS_TILE == BFQ_RS_TILE, "bfq_radix_block_elems() counts in scatter tiles");

template <class T>
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const T *__restrict__ dw, u64 n, int shift,
                                                           u32 *__restrict__ hist, u64 nblocks, u64 blockElems)
{
    __shared__ u32 wh[RS_WAVES][256];
    u32 w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&wh[0][0])[i] = 0;
    __syncthreads();
    u64 base = (u64)blockIdx.x * blockElems;
    u64 end = base + blockElems;
    if (end > n) end = n;
    // 16 bytes per lane and load (4 or 2 records); the block base is a multiple of 4 records
    constexpr int V = 16 / sizeof(T);
    u64 i = base + (u64)threadIdx.x * V;
    for (; i + V <= end; i += (u64)RS_THREADS * V) {
        uint4 x = *(const uint4 *)(dw + i);
        if (sizeof(T) == 4) {
            atomicAdd(&wh[w][(x.x >> shift) & 255u], 1u); atomicAdd(&wh[w][(x.y >> shift) & 255u], 1u);
            atomicAdd(&wh[w][(x.z >> shift) & 255u], 1u); atomicAdd(&wh[w][(x.w >> shift) & 255u], 1u);
        } else {                                                  // digit in the low word of each u64
            atomicAdd(&wh[w][(x.x >> shift) & 255u], 1u); atomicAdd(&wh[w][(x.z >> shift) & 255u], 1u);
        }
    }
    for (; i < end; i++) {                                        // at most V-1 records of the last block
        u32 d = (u32)(dw[i] >> shift) & 255u;
        atomicAdd(&wh[w][d], 1u);
    }
    __syncthreads();
    u32 d = threadIdx.x;
    if (d < 256) {
        u32 t = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) t += wh[k][d];
        hist[(u64)d * nblocks + blockIdx.x] = t;
    }
}

// DW = which word carries the digit of this pass (0: w0, 1: w1); 2: w1, and the records do not exist yet -- record i is made
// here from the text (suffix i: its 16-symbol key, position, preceding symbol and quality, exactly what k_build_keys
// would have written): the sort's first pass then reads 2.4 bytes per row instead of 12, and nobody writes those 12.
template <int DW>
__global__ __launch_bounds__(RS_THREADS, RS_OCC) void k_radix_scatter(SortRec in, SortRec out, u64 n, int shift,
                                                              const u64 *__restrict__ blockOff, u64 nblocks, u32 tilesPerBlock, u64 blockMul,
                                                              RadixText tx)
{
    __shared__ u64 stage[RS_TILE];          // w0, then the (w1,w2) pair of the records
    __shared__ u8 dig[RS_TILE];             // digit of every tile-sorted slot
    __shared__ u32 wcnt[RS_WAVES][256];     // per-wave digit counters -> exclusive prefix across waves
    __shared__ u32 lstart[256];             // first tile-sorted slot of each digit
    __shared__ u64 gbase[256];              // global output cursor of each digit for this workgroup
    __shared__ u32 shscan[RS_WAVES];

    const u32 tid = threadIdx.x, lane = bfq_lane(), w = tid >> 6;
    const u64 ltmask = bfq_lanemask_lt();
    // Which block this workgroup sorts: in order, or (blockMul != 0, BFQ_RS_PERM=1) spread by a stride coprime to the block
    // count, so that the ~1000 resident workgroups write all over every digit's bucket instead of into 256 windows of a few
    // MB that move in lock-step.  Built to test whether such windows explain why a pass costs 26 or 34 ms depending on where
    // the buffers lie: they do not -- the times are the same either way (profiles/r3/placement.md).
    const u64 blk = blockMul ? ((u64)blockIdx.x * blockMul) % nblocks : (u64)blockIdx.x;
    if (tid < 256) gbase[tid] = blockOff[(u64)tid * nblocks + blk];

    for (u32 t = 0; t < tilesPerBlock; t++) {
        u64 tbase = (blk * tilesPerBlock + t) * RS_TILE;
        if (tbase >= n) break;                                   // uniform
        u32 cnt = (n - tbase < (u64)RS_TILE) ? (u32)(n - tbase) : (u32)RS_TILE;

        for (int i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&wcnt[0][0])[i] = 0;
        __syncthreads();

        // tile order = (wave, round, lane): wave w owns slots [w*1024, w*1024+1024)
        u32 a0[RS_ROUNDS], a1[RS_ROUNDS], a2[RS_ROUNDS];
        u32 pk[RS_ROUNDS];                                       // digit << 16 | rank, later digit << 16 | tile slot
        // unconditional loads (slots past the end of the last tile re-read its last record), padding selected afterwards:
        // all 24 loads of a lane are in flight together
        const u32 lastSlot = cnt - 1;
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++) {
            const u32 slot = w * (RS_ROUNDS * 64) + r * 64 + lane;
            const u64 src = tbase + (slot < lastSlot ? slot : lastSlot);
            if (DW == 2) {
                const u64 wd = src / BFQ_SYMS_PER_WORD;
                const u32 o3 = (u32)(src - wd * BFQ_SYMS_PER_WORD) * 3u;
                const u64 *t3 = tx.text3 + bfq_t3_at(wd);
                const u64 hi = (t3[0] << o3) & BFQ_M63;
                const u64 lo = o3 ? (t3[1] >> (63u - o3)) : 0ull;
                const u64 sk = bfq_skey_of(bfq_mask_key(hi | lo));
                const u32 pc = src ? (u32)tx.T8[src - 1] : 0u;
                const u32 pq = pc ? (u32)tx.Q8[src - 1] : (u32)'#';
                const u64 pay = bfq_pack_val(src, pc, pq);
                a0[r] = bfq_rec_w0(sk); a1[r] = bfq_rec_w1(sk, pay); a2[r] = bfq_rec_w2(pay);
            } else {
                a0[r] = in.w0[src];
                const u64 x12 = in.w12[src];
                a1[r] = (u32)x12;
                a2[r] = (u32)(x12 >> 32);
            }
        }
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++) {
            const u32 slot = w * (RS_ROUNDS * 64) + r * 64 + lane;
            if (slot >= cnt) { a0[r] = 0xFFFFFFFFu; a1[r] = 0xFFFFFFFFu; a2[r] = 0u; }   // padding sorts last (digit 255, tile end)
        }
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++) {
            u32 d = ((DW ? a1[r] : a0[r]) >> shift) & 255u;
            u64 peers = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                bool bit = (d >> b) & 1u;
                u64 bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            u32 before = (u32)__popcll(peers & ltmask);
            u32 c0 = wcnt[w][d];
            pk[r] = (d << 16) | (c0 + before);
            bfq_wave_sync();
            if (before == 0) wcnt[w][d] = c0 + (u32)__popcll(peers);
            bfq_wave_sync();
        }
        __syncthreads();

        // per digit (thread = digit): prefix across waves, then across digits
        u32 tot = 0;
        if (tid < 256) {
#pragma unroll
            for (int k = 0; k < RS_WAVES; k++) { u32 ck = wcnt[k][tid]; wcnt[k][tid] = tot; tot += ck; }
        }
        {                                                        // exclusive scan of tot over the workgroup
            u32 inc = bfq_wave_incscan32_bp(tot);
            if (lane == 63) shscan[w] = inc;
            __syncthreads();
            u32 base = 0;
#pragma unroll
            for (int k = 0; k < RS_WAVES; k++) base += (k < (int)w) ? shscan[k] : 0u;
            if (tid < 256) lstart[tid] = base + inc - tot;
        }
        __syncthreads();

#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++) {
            u32 d = pk[r] >> 16;
            u32 p = lstart[d] + wcnt[w][d] + (pk[r] & 0xFFFFu);
            pk[r] = p;
            stage[p] = (u64)a0[r];
            dig[p] = (u8)d;
        }
        __syncthreads();
        u64 dst[RS_ROUNDS];
#pragma unroll
        for (int q = 0; q < RS_ROUNDS; q++) {
            u32 j = q * RS_THREADS + tid;
            u32 d = dig[j];
            dst[q] = gbase[d] + (u64)(j - lstart[d]);
            if (j < cnt) out.w0[dst[q]] = (u32)stage[j];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++) stage[pk[r]] = ((u64)a2[r] << 32) | a1[r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RS_ROUNDS; q++) {
            u32 j = q * RS_THREADS + tid;
            if (j < cnt) out.w12[dst[q]] = stage[j];
        }
        __syncthreads();
        if (tid < 256) gbase[tid] += tot;
        // next iteration starts with a barrier after zeroing wcnt
    }
}

// skey digits: 0 in w1 (bits 24..31), 1..4 in w0.  The records start in `in`; `tmp` is the other ping-pong buffer.
// Returns the buffer that holds the result: `in` for an even number of passes, `tmp` for an odd one.
SortRec bfq_radix_sort(bfq_ctx *c, SortRec in, SortRec tmp, u64 n, int passes, const u32 *hist0, const RadixText *fromText)
{
    if (n < 2) {
        if (n == 1 && (passes & 1)) {
            HIP_CHECK(hipMemcpyAsync(tmp.w0, in.w0, 4, hipMemcpyDeviceToDevice, c->stream));
            HIP_CHECK(hipMemcpyAsync(tmp.w12, in.w12, 8, hipMemcpyDeviceToDevice, c->stream));
        }
        return (passes & 1) ? tmp : in;
    }
    const u64 be = bfq_radix_block_elems(n);
    const u32 tpb = (u32)(be / RS_TILE);
    u64 nb = ceil_div(n, be);
    size_t m = c->mark();
    u32 *hist = c->alloc<u32>(256 * nb);
    u64 *off = c->alloc<u64>(256 * nb);
    SortRec out = tmp;
    // stride of the block permutation: near the golden section of the block count, coprime to it (0: blocks in order)
    u64 mul = 0;
    if (c->env.rsPerm && nb >= 64) {
        auto gcd = [](u64 a, u64 b) { while (b) { const u64 t = a % b; a = b; b = t; } return a; };
        mul = (u64)((double)nb * 0.6180339887) | 1ull;
        while (gcd(mul, nb) != 1) mul += 2;
    }
    for (int pass = 0; pass < passes; pass++) {
        const int dw = pass == 0 ? 1 : 0;
        const int shift = pass == 0 ? 24 : 8 * (pass - 1);
        const bool have = (pass == 0 && hist0);            // pass 0's counts were made by k_build_keys
        if (!have && dw)
            KLAUNCH(c, K_RADIX_HIST, 8.0 * (double)n, k_radix_hist<u64>, nb, RS_THREADS, (const u64 *)in.w12, n, shift, hist, nb, be);
        else if (!have)
            KLAUNCH(c, K_RADIX_HIST, 4.0 * (double)n, k_radix_hist<u32>, nb, RS_THREADS, (const u32 *)in.w0, n, shift, hist, nb, be);
        bfq_exscan_u32(c, have ? hist0 : hist, off, 256 * nb, nullptr);
        const RadixText none{nullptr, nullptr, nullptr};
        if (dw && fromText && have)                        // the records are made on the way: 12 B written + 2.4 B of text read per row
            KLAUNCH(c, K_RADIX_SCATTER, 14.4 * (double)n, k_radix_scatter<2>, nb, RS_THREADS, in, out, n, shift, (const u64 *)off, nb, tpb, mul, *fromText);
        else if (dw)
            KLAUNCH(c, K_RADIX_SCATTER, 24.0 * (double)n, k_radix_scatter<1>, nb, RS_THREADS, in, out, n, shift, (const u64 *)off, nb, tpb, mul, none);
        else
            KLAUNCH(c, K_RADIX_SCATTER, 24.0 * (double)n, k_radix_scatter<0>, nb, RS_THREADS, in, out, n, shift, (const u64 *)off, nb, tpb, mul, none);
        SortRec t = in; in = out; out = t;
    }
    c->release(m);
    return in;
}
