// k_rank.hip -- builds the LF table (bfq_rank.h) from the eBWT bytes and the permuted
// qualities: the batched form of the reference's rank queries.  Replaces dna_string_n
// construction + build_rank_support (external/bwt2lcp/dna_string_n.hpp:52-109,247-285),
// the F-array loop of dna_bwt_n.hpp:46-61, LF() of dna_bwt_n.hpp:80-101 and the QUAL
// array load of bfq_int.cpp:640-651.  Forbidden symbols raise errSymbol
// (dna_string_n.hpp:87-93 exits 1).
//   k_lf_count : symbol counts of every 256-row group (skipped when k_emit_bwt already made them)
//   scans      : counts -> occurrences before each group, totals -> F array
//   k_lf_build : per row F[c] + before-group + before-row-in-group (packed counters, wave
//                scan), packed with code and quality                (2 B read, 8 B written)
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_rank.h"

// One wavefront per group of 256 rows, 4 consecutive rows per lane (one 4-byte load of the eBWT and
// one of the qualities).  Per-lane symbol counts travel packed in a u64 (6 fields of 10 bits).
__device__ __forceinline__ u32 byte_code(u32 ch, u32 term, bool *bad)
{
    u32 code = (ch == term) ? 0u : bfq_base_code((u8)ch);
    if (code == BFQ_CODE_INVALID) { *bad = true; code = 4; }
    return code;
}
// codes of rows r0..r0+3 (7 = past the end) from 4 eBWT bytes
__device__ __forceinline__ void codes4(const u8 *__restrict__ bwt, u64 r0, u64 n, u32 term, u32 *code, bool *bad)
{
    u32 x = 0;
    if (r0 + 4 <= n) x = *(const u32 *)(bwt + r0);
    else for (u64 k = 0; r0 + k < n; k++) x |= (u32)bwt[r0 + k] << (8 * k);
#pragma unroll
    for (int k = 0; k < 4; k++) code[k] = (r0 + k < n) ? byte_code((x >> (8 * k)) & 0xFFu, term, bad) : 7u;
}
__device__ __forceinline__ u64 packed_counts(const u32 *code)
{
    u64 p = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (code[k] < 6) p += 1ull << (10 * code[k]);
    return p;
}

__global__ __launch_bounds__(256) void k_lf_count(const u8 *__restrict__ bwt, u64 n, u32 term, u32 *__restrict__ gcnt,
                                                  u64 ngroups, DevCounters *cnt)
{
    const u32 lane = bfq_lane();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    bool bad = false;
    for (u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; g < ngroups; g += nwaves) {
        u32 code[4];
        codes4(bwt, g * 256 + (u64)lane * 4, n, term, code, &bad);
        u64 p = packed_counts(code);
        p = bfq_readlane64(bfq_wave_incscan64(p), 63);                                  // wave total in every lane
        if (lane < 6) gcnt[(u64)lane * ngroups + g] = (u32)(p >> (10 * lane)) & 0x3FFu;
    }
    if (bad) atomicAdd(&cnt->errSymbol, 1ull);
}

__global__ __launch_bounds__(256) void k_lf_build(const u8 *__restrict__ bwt, const u8 *__restrict__ qs, u64 n, u32 term,
                                                  const u64 *__restrict__ scanned, u64 ngroups, const DevCounters *cnt,
                                                  u64 *__restrict__ lfq)
{
    const u32 lane = bfq_lane();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    u64 F[6];                                       // F array in order # A C G N T (dna_bwt_n.hpp:46-61)
    {
        u64 acc = 0;
#pragma unroll
        for (int s = 0; s < 6; s++) { F[s] = acc; acc += cnt->tot[s]; }
    }
    bool bad = false;
    for (u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; g < ngroups; g += nwaves) {
        const u64 r0 = g * 256 + (u64)lane * 4;
        u32 code[4];
        codes4(bwt, r0, n, term, code, &bad);
        u32 q4 = 0;
        if (r0 + 4 <= n) q4 = *(const u32 *)(qs + r0);
        else for (u64 k = 0; r0 + k < n; k++) q4 |= (u32)qs[r0 + k] << (8 * k);
        u64 p = packed_counts(code);
        u64 ex = bfq_wave_incscan64(p) - p;         // packed counts of the rows before mine in the group
        // F[c] + occurrences of c before this group: loaded by lanes 1..5, handed to everyone as wave-uniform values
        u64 fg = (lane >= 1 && lane <= 5) ? scanned[(u64)lane * ngroups + g] : 0ull;
        const u64 G1 = F[1] + bfq_readlane64(fg, 1), G2 = F[2] + bfq_readlane64(fg, 2), G3 = F[3] + bfq_readlane64(fg, 3),
                  G4 = F[4] + bfq_readlane64(fg, 4), G5 = F[5] + bfq_readlane64(fg, 5);
        u64 outv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 c = code[k];
            u64 x = 0;
            if (c >= 1 && c <= 5) {
                u32 before = (u32)(ex >> (10 * c)) & 0x3FFu;
                x = (c == 1 ? G1 : c == 2 ? G2 : c == 3 ? G3 : c == 4 ? G4 : G5) + before;
            }
            if (c < 6) ex += 1ull << (10 * c);
            outv[k] = x | ((u64)(c & 7u) << 40) | ((u64)((q4 >> (8 * k)) & 0xFFu) << 48);
        }
        if (r0 + 4 <= n) {
            *(ulonglong2 *)(lfq + r0) = make_ulonglong2(outv[0], outv[1]);
            *(ulonglong2 *)(lfq + r0 + 2) = make_ulonglong2(outv[2], outv[3]);
        } else {
            for (u64 k = 0; r0 + k < n; k++) lfq[r0 + k] = outv[k];
        }
    }
}

// occurrences of every symbol before each 256-row group ([6][ngroups], arena) + the totals in d_cnt->tot
u64 *bfq_symbol_scans(bfq_ctx *c, const u8 *bwt, u64 n, int term, const u32 *gcntIn, u32 *gcntOut)
{
    u64 ngroups = n / 256 + 1;
    u64 *scanned = c->alloc<u64>(6 * ngroups);
    size_t m = c->mark();
    u32 *gcnt = gcntIn ? const_cast<u32 *>(gcntIn) : gcntOut ? gcntOut : c->alloc<u32>(6 * ngroups);
    if (!gcntIn) KLAUNCH(c, K_RANK_BUILD, (double)n, k_lf_count, bfq_grid(ngroups, 4), 256, bwt, n, (u32)(term & 0xFF), gcnt, ngroups, c->d_cnt);
    for (int s = 0; s < 6; s++) bfq_exscan_u32(c, gcnt + (u64)s * ngroups, scanned + (u64)s * ngroups, ngroups, &c->d_cnt->tot[s]);
    c->release(m);
    return scanned;
}

RankIndex bfq_rank_build(bfq_ctx *c, const u8 *bwt, const u8 *qs, u64 n, int term, const u32 *gcntIn)
{
    RankIndex R;
    u64 ngroups = n / 256 + 1;
    u64 *lfq = c->alloc<u64>(n + 8);
    size_t m = c->mark();
    u64 *scanned = bfq_symbol_scans(c, bwt, n, term, gcntIn, nullptr);
    KLAUNCH(c, K_RANK_FINAL, 10.0 * (double)n, k_lf_build, bfq_grid(ngroups, 4), 256, bwt, qs, n, (u32)(term & 0xFF),
            (const u64 *)scanned, ngroups, (const DevCounters *)c->d_cnt, lfq);
    c->release(m);
    R.lfq = lfq; R.n = n;
    return R;
}
