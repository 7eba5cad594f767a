// k_radix.hip -- stable LSD radix sort of (63-bit key, 64-bit payload) pairs, 8-bit digits.
//
// This is the sort at the heart of the eBWT construction that replaces
// `gsufsort --bwt --qs` (call site BFQzip.py:184): one pair per read suffix,
// key = the suffix's first 21 symbols.  Per pass:
//   k_radix_hist    : per-workgroup digit counts (per-wave LDS histograms)
//   exclusive scan  : digit-major table -> global offsets (k_scan.hip)
//   k_radix_scatter : wave-level match ranking (ballots), per-wave LDS digit
//                     counters + prefix across waves, pairs staged in LDS in
//                     tile-sorted order, then written out in coalesced runs.
// HBM-bound integer work: per pass 8 B/key read (hist) + 16 B read + 16 B written.
#include "bfq_internal.h"
#include "bfq_device.h"

#define RS_THREADS 256
#define RS_WAVES 4
#define RS_ROUNDS 12                                // items per thread
#define RS_TILE (RS_THREADS * RS_ROUNDS)            // 4096 pairs per tile
#define RS_TILES_PER_BLOCK 8
#define RS_BLOCK_ELEMS ((u64)RS_TILE * RS_TILES_PER_BLOCK)

__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const u64 *__restrict__ keys, u64 n, int shift,
                                                           u32 *__restrict__ hist, u64 nblocks)
{
    __shared__ u32 wh[RS_WAVES][256];
    u32 w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&wh[0][0])[i] = 0;
    __syncthreads();
    u64 base = (u64)blockIdx.x * RS_BLOCK_ELEMS;
    u64 end = base + RS_BLOCK_ELEMS;
    if (end > n) end = n;
    for (u64 i = base + threadIdx.x; i < end; i += RS_THREADS) {
        u32 d = (u32)(keys[i] >> shift) & 255u;
        atomicAdd(&wh[w][d], 1u);
    }
    __syncthreads();
    u32 d = threadIdx.x;
    hist[(u64)d * nblocks + blockIdx.x] = wh[0][d] + wh[1][d] + wh[2][d] + wh[3][d];
}

__global__ __launch_bounds__(RS_THREADS, 4) void k_radix_scatter(const u64 *__restrict__ kin, const u64 *__restrict__ vin,
                                                              u64 *__restrict__ kout, u64 *__restrict__ vout, u64 n,
                                                              int shift, const u64 *__restrict__ blockOff, u64 nblocks)
{
    __shared__ u64 stage[RS_TILE];          // 32 KiB: keys, then payloads
    __shared__ u32 wcnt[RS_WAVES][256];     // per-wave digit counters -> exclusive prefix across waves
    __shared__ u32 lstart[256];             // first tile-sorted slot of each digit
    __shared__ u64 gbase[256];              // global output cursor of each digit for this workgroup
    __shared__ u32 shscan[4];

    const u32 tid = threadIdx.x, lane = bfq_lane(), w = tid >> 6;
    const u64 ltmask = bfq_lanemask_lt();
    gbase[tid] = blockOff[(u64)tid * nblocks + blockIdx.x];

    for (int t = 0; t < RS_TILES_PER_BLOCK; t++) {
        u64 tbase = (u64)blockIdx.x * RS_BLOCK_ELEMS + (u64)t * RS_TILE;
        if (tbase >= n) break;                                   // uniform
        u32 cnt = (n - tbase < (u64)RS_TILE) ? (u32)(n - tbase) : (u32)RS_TILE;

        for (int i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&wcnt[0][0])[i] = 0;
        __syncthreads();

        // tile order = (wave, round, lane): wave w owns slots [w*1024, w*1024+1024)
        u64 k[RS_ROUNDS];
        u32 pk[RS_ROUNDS];                                       // digit << 16 | rank, later digit << 16 | tile slot
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++) {
            u32 slot = w * (RS_ROUNDS * 64) + r * 64 + lane;
            k[r] = (slot < cnt) ? kin[tbase + slot] : ~0ull;     // padding sorts last (digit 255, tile end)
        }
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++) {
            u32 d = (u32)(k[r] >> shift) & 255u;
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
            __builtin_amdgcn_wave_barrier();
            if (before == 0) wcnt[w][d] = c0 + (u32)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();

        // per digit (thread = digit): prefix across waves, then across digits
        u32 c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid], c3 = wcnt[3][tid];
        u32 tot = c0 + c1 + c2 + c3, dummy;
        wcnt[0][tid] = 0; wcnt[1][tid] = c0; wcnt[2][tid] = c0 + c1; wcnt[3][tid] = c0 + c1 + c2;
        u32 ls = bfq_block_exscan32(tot, shscan, &dummy);        // two barriers inside
        lstart[tid] = ls;
        __syncthreads();

#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++) {
            u32 d = pk[r] >> 16;
            u32 p = lstart[d] + wcnt[w][d] + (pk[r] & 0xFFFFu);
            pk[r] = (d << 16) | p;
            stage[p] = k[r];
        }
        // payloads are requested now so that their latency overlaps the key write-out
        u64 v[RS_ROUNDS];
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++) {
            u32 slot = w * (RS_ROUNDS * 64) + r * 64 + lane;
            v[r] = (slot < cnt) ? vin[tbase + slot] : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RS_ROUNDS; q++) {
            u32 j = q * RS_THREADS + tid;
            u64 key = stage[j];
            u32 d = (u32)(key >> shift) & 255u;
            if (j < cnt) kout[gbase[d] + (u64)(j - lstart[d])] = key;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; r++)                      // the payload's free top byte carries the digit
            stage[pk[r] & 0xFFFFu] = v[r] | ((u64)(pk[r] >> 16) << 56);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RS_ROUNDS; q++) {
            u32 j = q * RS_THREADS + tid;
            u64 x = stage[j];
            u32 d = (u32)(x >> 56);
            if (j < cnt) vout[gbase[d] + (u64)(j - lstart[d])] = x & 0x00FFFFFFFFFFFFFFull;
        }
        __syncthreads();
        gbase[tid] += tot;
        // next iteration starts with a barrier after zeroing wcnt
    }
}

// keySyms = 21: all 63 key bits (8 passes); keySyms = 16: only the first 16 symbols = bits 15..62
// (6 passes) -- rows equal on them stay in position order and are finished by the refinement.
void bfq_radix_sort(bfq_ctx *c, u64 *keysA, u64 *valsA, u64 *keysB, u64 *valsB, u64 n, int keySyms)
{
    if (n < 2) return;
    u64 nb = ceil_div(n, RS_BLOCK_ELEMS);
    size_t m = c->mark();
    u32 *hist = c->alloc<u32>(256 * nb);
    u64 *off = c->alloc<u64>(256 * nb);
    u64 *kin = keysA, *vin = valsA, *kout = keysB, *vout = valsB;
    const int lowbit = 3 * (BFQ_SYMS_PER_WORD - keySyms);       // 0 or 15
    const int npass = (63 - lowbit + 7) / 8;                     // 8 or 6: even, so the result returns to A
    for (int pass = 0; pass < npass; pass++) {
        int shift = lowbit + pass * 8;
        KLAUNCH(c, K_RADIX_HIST, 8.0 * (double)n, k_radix_hist, nb, RS_THREADS, (const u64 *)kin, n, shift, hist, nb);
        bfq_exscan_u32(c, hist, off, 256 * nb, nullptr);
        KLAUNCH(c, K_RADIX_SCATTER, 32.0 * (double)n, k_radix_scatter, nb, RS_THREADS, (const u64 *)kin,
                (const u64 *)vin, kout, vout, n, shift, (const u64 *)off, nb);
        u64 *t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    // an even number of passes: result is back in keysA / valsA
    c->release(m);
}
