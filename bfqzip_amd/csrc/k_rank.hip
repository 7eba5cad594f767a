// k_rank.hip -- builds the LF table (bfq_rank.h) from the eBWT bytes and the permuted
// qualities: the batched form of the reference's rank queries.  Replaces dna_string_n
// construction + build_rank_support (external/bwt2lcp/dna_string_n.hpp:52-109,247-285),
// the F-array loop of dna_bwt_n.hpp:46-61, LF() of dna_bwt_n.hpp:80-101 and the QUAL
// array load of bfq_int.cpp:640-651.  Forbidden symbols raise errSymbol
// (dna_string_n.hpp:87-93 exits 1).
//   k_lf_count : symbol counts of every 256-row group            (1 B/row read)
//   scans      : counts -> occurrences before each group, totals -> F array
//   k_lf_build : per row F[c] + before-group + before-row-in-group (ballot bit vectors,
//                popcounts), packed with code and quality          (2 B read, 8 B written)
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_rank.h"

__device__ __forceinline__ u32 row_code(const u8 *__restrict__ bwt, u64 r, u64 n, u32 term, DevCounters *cnt)
{
    if (r >= n) return 7u;                          // rows past the end match no symbol
    u8 ch = bwt[r];
    u32 code = (ch == (u8)term) ? 0u : bfq_base_code(ch);
    if (code == BFQ_CODE_INVALID) { if (cnt) atomicAdd(&cnt->errSymbol, 1ull); code = 4; }
    return code;
}

__global__ __launch_bounds__(256) void k_lf_count(const u8 *__restrict__ bwt, u64 n, u32 term, u32 *__restrict__ gcnt,
                                                  u64 ngroups, DevCounters *cnt)
{
    __shared__ u32 wc[4][6];
    const u32 lane = bfq_lane(), w = threadIdx.x >> 6;
    for (u64 g = blockIdx.x; g < ngroups; g += gridDim.x) {
        u32 code = row_code(bwt, g * 256 + threadIdx.x, n, term, cnt);
        u64 b0 = __ballot(code & 1u), b1 = __ballot(code & 2u), b2 = __ballot(code & 4u);
        if (lane < 6) {
            u64 m = ((lane & 1u) ? b0 : ~b0) & ((lane & 2u) ? b1 : ~b1) & ((lane & 4u) ? b2 : ~b2);
            wc[w][lane] = (u32)__popcll(m);
        }
        __syncthreads();
        if (threadIdx.x < 6) gcnt[(u64)threadIdx.x * ngroups + g] = wc[0][threadIdx.x] + wc[1][threadIdx.x] + wc[2][threadIdx.x] + wc[3][threadIdx.x];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_lf_build(const u8 *__restrict__ bwt, const u8 *__restrict__ qs, u64 n, u32 term,
                                                  const u64 *__restrict__ scanned, u64 ngroups, const DevCounters *cnt,
                                                  u64 *__restrict__ lfq)
{
    __shared__ u32 wc[4][6];
    __shared__ u64 F[6];
    const u32 lane = bfq_lane(), w = threadIdx.x >> 6;
    if (threadIdx.x == 0) {                         // F array in order # A C G N T (dna_bwt_n.hpp:46-61)
        u64 acc = 0;
        for (int s = 0; s < 6; s++) { F[s] = acc; acc += cnt->tot[s]; }
    }
    const u64 ltmask = bfq_lanemask_lt();
    for (u64 g = blockIdx.x; g < ngroups; g += gridDim.x) {
        u64 r = g * 256 + threadIdx.x;
        u32 code = row_code(bwt, r, n, term, nullptr);
        u64 b0 = __ballot(code & 1u), b1 = __ballot(code & 2u), b2 = __ballot(code & 4u);
        u64 peers = ((code & 1u) ? b0 : ~b0) & ((code & 2u) ? b1 : ~b1) & ((code & 4u) ? b2 : ~b2);
        if (lane < 6) {
            u64 m = ((lane & 1u) ? b0 : ~b0) & ((lane & 2u) ? b1 : ~b1) & ((lane & 4u) ? b2 : ~b2);
            wc[w][lane] = (u32)__popcll(m);
        }
        __syncthreads();                            // also orders the F[] initialisation
        if (r < n) {
            u64 x = 0;
            if (code >= 1 && code <= 5) {
                u32 before = (u32)__popcll(peers & ltmask);
                for (u32 k = 0; k < w; k++) before += wc[k][code];
                x = F[code] + scanned[(u64)code * ngroups + g] + before;
            }
            lfq[r] = x | ((u64)code << 40) | ((u64)qs[r] << 48);
        }
        __syncthreads();
    }
}

RankIndex bfq_rank_build(bfq_ctx *c, const u8 *bwt, const u8 *qs, u64 n, int term)
{
    RankIndex R;
    u64 ngroups = n / 256 + 1;
    u64 *lfq = c->alloc<u64>(n + 8);
    size_t m = c->mark();
    u32 *gcnt = c->alloc<u32>(6 * ngroups);
    u64 *scanned = c->alloc<u64>(6 * ngroups);
    KLAUNCH(c, K_RANK_BUILD, (double)n, k_lf_count, bfq_grid(ngroups, 1), 256, bwt, n, (u32)(term & 0xFF), gcnt, ngroups, c->d_cnt);
    for (int s = 0; s < 6; s++) bfq_exscan_u32(c, gcnt + (u64)s * ngroups, scanned + (u64)s * ngroups, ngroups, &c->d_cnt->tot[s]);
    KLAUNCH(c, K_RANK_FINAL, 10.0 * (double)n, k_lf_build, bfq_grid(ngroups, 1), 256, bwt, qs, n, (u32)(term & 0xFF),
            (const u64 *)scanned, ngroups, (const DevCounters *)c->d_cnt, lfq);
    c->release(m);
    R.lfq = lfq; R.n = n;
    return R;
}
