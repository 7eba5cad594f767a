// k_compact.hip -- steps 2-4 on a GIVEN eBWT under a workspace cap (bfq_int / bfq_ext: SURVEY rows a6-a11, 8(f).2).
//
// The fast path tabulates every rank query in an 8-byte LF entry per row (bfq_rank.h): 17 bytes per row of workspace with
// the flags and the outputs.  Below that (bfq_params.ws_cap_mib / BFQ_WS_CAP) the same results come from the reference's
// own arrangement, on the GPU: a succinct rank structure answered on demand (the 64-byte rank blocks of k_bfs.hip, 1 byte
// per row: dna_string_n.hpp:112-185,367-406; LF, dna_bwt_n.hpp:80-101), the qualities smoothed in place (QUAL[],
// bfq_int.cpp:386-405) and the replaced bases in a side array (rankbv + BWT_MOD, bfq_int.cpp:386-391,782).  An LF step of
// the inversion then costs a rank block, a quality byte and a replacement byte instead of one table entry -- about three
// sectors instead of one -- for 5 n bytes less:
//   eBWT n + qualities n (caller's buffers) | rank blocks n | in() flags n | replacements n | reads out 2 (n - N)
// The LCP never exists as a whole: an LCP file (bfq_ext) is streamed through a window and turned into flags chunk by chunk.
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_rankblk.h"

struct CompactIndex { const u64 *rank; u64 F[6]; const u8 *qual; const u8 *repl; u64 n; };

__global__ __launch_bounds__(256) void k_invert_count_rank(CompactIndex R, u64 N, u32 *__restrict__ lens, DevCounters *cnt)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        u64 j = i;
        u32 len = 0;
        for (;;) {
            const RankBlk b = load_blk(R.rank, j >> 6);
            const u32 code = blk_code(b, j);
            if (!code) break;
            const u64 nx = R.F[code] + blk_occ_of(b, j, code);
            if (++len > BFQ_MAX_READ_LEN || nx >= R.n) { atomicAdd(&cnt->errInvert, 1ull); break; }
            j = nx;
        }
        lens[i] = len;
    }
}

// invert() of bfq_int.cpp:748-819, one walk per read, on the compact structure.  The read is produced back to front; 8 bytes
// of each stream are collected in a register and stored when full (unaligned 8-byte stores), the front of the read byte-wise.
// NL: line-stream layout (read i at roff[i] + i, then '\n').
template <int NL>
__global__ __launch_bounds__(256) void k_invert_rank(CompactIndex R, u64 first, u64 count, const u64 *__restrict__ roff, int B,
                                                     u8 *__restrict__ out_bases, u8 *__restrict__ out_quals, DevCounters *cnt)
{
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (u64)gridDim.x * blockDim.x) {
        const u64 i = first + t;
        const u64 lo = roff[i] + (NL ? i : 0), end = roff[i + 1] + (NL ? i : 0);
        if (NL) { out_bases[end] = 10; out_quals[end] = 10; }
        u64 pos = end, j = i;
        u64 bw = 0, qw = 0;
        u32 have = 0;
        bool bad = false;
        while (pos > lo) {
            const RankBlk b = load_blk(R.rank, j >> 6);
            const u32 code = blk_code(b, j);
            if (!code) { bad = true; break; }                          // walk ended before the read did
            const u64 nx = R.F[code] + blk_occ_of(b, j, code);
            if (nx >= R.n) { bad = true; break; }
            const u32 rp = __builtin_nontemporal_load(R.repl + j);
            const u32 sym = rp ? rp : (u32)bfq_code_sym(code);
            u32 q = __builtin_nontemporal_load(R.qual + j);
            if (B) q = bfq_bin8(q);
            --pos;
            bw = (bw << 8) | sym; qw = (qw << 8) | q;                  // the byte for the lowest address ends up in the low byte
            if (++have == 8) { *(u64 *)(out_bases + pos) = bw; *(u64 *)(out_quals + pos) = qw; bw = qw = 0; have = 0; }
            j = nx;
        }
        if (!bad) for (u32 k = 0; k < have; k++) { out_bases[lo + k] = (u8)(bw >> (8 * k)); out_quals[lo + k] = (u8)(qw >> (8 * k)); }
        if (!bad) { const RankBlk b = load_blk(R.rank, j >> 6); if (blk_code(b, j) != 0) bad = true; }   // read longer than its slot
        if (bad) atomicAdd(&cnt->errInvert, 1ull);
    }
}

// in() flags of rows [rbase, rbase + cnt) from a window of LCP values: win[t] = LCP[rbase - 1 + t] (win[0] unused for row 0)
__global__ __launch_bounds__(256) void k_lcp_flags_win(const u16 *__restrict__ win, u64 rbase, u64 cnt, u64 n, int K, u8 *__restrict__ in)
{
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < cnt; t += (u64)gridDim.x * blockDim.x) {
        const u64 r = rbase + t;
        const int lp = r >= 1 ? (int)win[t] : 0, l = (int)win[t + 1], ln = r + 1 < n ? (int)win[t + 2] : 0;
        const bool thr = r >= 1 && l >= K;
        const bool mn = r >= 1 && r + 2 <= n && lp > l && ln >= l;
        in[r] = (thr && !mn) ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void k_lcp_widen_win(const u8 *__restrict__ raw, int lb, u64 cnt, u16 *__restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (u64)gridDim.x * blockDim.x) {
        const u64 v = lb == 1 ? raw[i] : lb == 2 ? ((const u16 *)raw)[i] : ((const u32 *)raw)[i];
        out[i] = (u16)(v > 0xFFFE ? 0xFFFE : v);
    }
}

u64 *bfq_symbol_scans(bfq_ctx *c, const u8 *bwt, u64 n, int term, const u32 *gcntIn, u32 *gcntOut);   // k_rank.hip
void bfq_rank_blocks(bfq_ctx *c, const u8 *bwt, u64 n, int term, const u64 *scanned, u64 *rank);          // k_bfs.hip

static u64 lcp_window(bfq_ctx *c, u64 n)
{
    // (up to a third of the rows the window's 6 bytes per row lie where the reads go out later: it costs no workspace)
    u64 W = c->env.compactWin ? c->env.compactWin : BFQ_COMPACT_LCP_WIN;
    if (!c->env.compactWin && W > n / 3) W = n / 3 > (1u << 20) ? n / 3 : (1u << 20);
    if (W > n) W = n;
    return W ? W : 1;
}
// What the arena holds: rank blocks, flags, replacements (3 n) for the whole call; then either the LCP scratch (bfq_ext: a
// window of the LCP file; bfq_int: the 16-bit LCP, the refinement's ring queue and lists) or, once the flags exist, the
// reads going out (2 n).  The ring takes what the cap leaves (the caller's eBWT + qualities, `resident` bytes, count too):
// a sixteenth of the rows at least, the n entries of the plain log at most.
static size_t compact_fixed(u64 n, u64 N, u64 extra)
{
    return 3 * (n + 1024) + 72 * (n / 256 + 2) + 16 * (N + 64) + extra + (96u << 20);
}
static size_t compact_lcp_lists(u64 n) { return 2 * (n + 256) + 32 * (n / 65 + 64) + 64 * ((n >> 25) + 64) + 65536; }
static u64 ring_entries(bfq_ctx *c, u64 n, u64 N, u64 extra, size_t resident)
{
    if (c->env.compactRing) return c->env.compactRing;
    u64 lo = n / 16 > 65536 ? n / 16 : 65536, hi = n + 64;
    if (!c->wsLimit()) return hi;
    const size_t used = compact_fixed(n, N, extra) + compact_lcp_lists(n) + resident;
    const u64 fit = c->wsLimit() > used ? (c->wsLimit() - used) / 8 : 0;
    return fit < lo ? lo : fit > hi ? hi : fit;
}
size_t bfq_ws_need_compact(bfq_ctx *c, u64 n, u64 N, u64 extra, bool haveLcp, size_t resident)
{
    const size_t out = 2 * (n + 256);                           // reads out
    size_t lcp;
    if (haveLcp) lcp = (size_t)(lcp_window(c, n) + 4) * 6 + (n >> 20) * 64 + 4096;
    else lcp = compact_lcp_lists(n) + 8 * (size_t)ring_entries(c, n, N, extra, resident);
    return compact_fixed(n, N, extra) + (lcp > out ? lcp : out);
}

// bwt / qual: the given eBWT and its permuted qualities on the device (qual is edited in place); lcp: the LCP file / array
// (lcp_bytes 1, 2 or 4 per entry).  Leaves the reads in ob / oq (packed) and their offsets in d_roff.
void bfq_steps234_compact(bfq_ctx *c, const u8 *bwt, u8 *qual, HostRef lcp, int lcp_bytes, u64 n, u64 N, u64 *d_roff, u32 *lens,
                          u8 **obOut, u8 **oqOut, u64 extra, size_t resident)
{
    c->n = n; c->N = N;
    const u64 ngroups = n / 256 + 1;
    u64 *scanned = bfq_symbol_scans(c, bwt, n, c->P.term, nullptr, nullptr);
    u64 *rank = c->alloc<u64>((ngroups * 4 + 2) * 8);
    bfq_rank_blocks(c, bwt, n, c->P.term, scanned, rank);
    c->fetchCounters();                                            // symbol totals -> F
    CompactIndex R;
    {
        u64 acc = 0;
        for (int s = 0; s < 6; s++) { R.F[s] = acc; acc += c->h_cnt.tot[s]; }
        if (acc != n || c->h_cnt.tot[0] != N) throw BfqError{BFQ_E_NOT_EBWT, "symbol counts do not add up to the eBWT"};
    }
    if (c->h_cnt.errSymbol) return;                                // reported by the caller's check
    u8 *in = c->alloc<u8>(n + 64), *repl = c->alloc<u8>(n + 64);
    HIP_CHECK(hipMemsetAsync(repl, 0, n + 64, c->stream));
    HIP_CHECK(hipMemsetAsync(in + n, 0, 64, c->stream));
    if (lcp.null()) {
        // bfq_int: the LCP deduced from the BWT alone (k_bfs.hip) on the same rank blocks, its queue a ring; the 16-bit array
        // lives only until the flags are made
        const size_t mk = c->mark();
        u16 *lcp16 = c->alloc<u16>(n + 64);
        bfq_lcp_from_bwt(c, bwt, n, N, c->P.term & 0xFF, lcp16, nullptr, rank, ring_entries(c, n, N, extra, resident));
        bfq_lcp_flags(c, lcp16, n, c->P.K, in);
        c->sync();
        c->release(mk);
    } else {   // the LCP file through a window: raw entries of rows [rb - 1, re + 1) -> 16 bits -> in() of rows [rb, re)
        const size_t mk = c->mark();
        const u64 W = lcp_window(c, n);
        u8 *raw = c->alloc<u8>((W + 4) * (size_t)lcp_bytes + 64);
        u16 *win = c->alloc<u16>(W + 4);
        for (u64 rb = 0; rb < n; rb += W) {
            const u64 re = rb + W < n ? rb + W : n;
            const u64 f0 = rb ? rb - 1 : 0, f1 = re + 1 < n ? re + 1 : n;       // entries fetched
            HostRef src = lcp;
            if (src.ptr) src.ptr = (char *)src.ptr + f0 * (u64)lcp_bytes; else src.off += f0 * (u64)lcp_bytes;
            bfq_upload(c, raw, src, (size_t)(f1 - f0) * (size_t)lcp_bytes);
            u16 *w0 = win + (rb ? 0 : 1);                                       // win[t] = LCP[rb - 1 + t]
            KLAUNCH(c, K_MISC, (double)(lcp_bytes + 2) * (double)(f1 - f0), k_lcp_widen_win, bfq_grid(f1 - f0, 256), 256, (const u8 *)raw, lcp_bytes, f1 - f0, w0);
            KLAUNCH(c, K_LCP_FLAGS, 3.0 * (double)(re - rb), k_lcp_flags_win, bfq_grid(re - rb, 256), 256, (const u16 *)win, rb, re - rb, n, c->P.K, in);
            c->sync();                                                         // the window is reused
        }
        c->release(mk);
    }
    u8 *ob = c->alloc<u8>(n - N + 64), *oq = c->alloc<u8>(n - N + 64);   // (where the LCP scratch was)
    *obOut = ob; *oqOut = oq;
    ClusterRank rm;
    rm.rankBlk = rank; rm.qual = qual; rm.repl = repl;
    for (int s = 0; s < 6; s++) rm.F[s] = R.F[s];
    RankIndex none{nullptr, n};
    bfq_clusters(c, none, bwt, qual, in, n, nullptr, &rm);
    R.rank = rank; R.qual = qual; R.repl = repl; R.n = n;
    bool guessed = false;
    auto count = [&] {
        if (N) KLAUNCH(c, K_INVERT_COUNT, 66.0 * (double)(n - N), k_invert_count_rank, bfq_grid(N, 256), 256, R, N, lens, c->d_cnt);
        bfq_exscan_u32(c, lens, d_roff, N, d_roff + N);
        u64 tot2 = 0;
        HIP_CHECK(hipMemcpyAsync(&tot2, d_roff + N, 8, hipMemcpyDeviceToHost, c->stream));
        c->fetchCounters();
        if (c->h_cnt.errInvert) throw BfqError{BFQ_E_NOT_EBWT, "LF walk did not close: not an eBWT of a read collection"};
        if (tot2 != n - N) throw BfqError{BFQ_E_NOT_EBWT, "LF walks do not cover the eBWT"};
    };
    if (N && (n - N) % N == 0 && !c->env.noLengthGuess) { bfq_fixed_offsets(c, N, (n - N) / N, d_roff); guessed = true; }
    else count();
    auto walk = [&] {
        if (N) KLAUNCH(c, K_INVERT, 68.0 * (double)(n - N), k_invert_rank<0>, bfq_grid(N, 256), 256, R, 0ull, N, (const u64 *)d_roff, c->P.B, ob, oq, c->d_cnt);
    };
    walk();
    if (guessed) {
        c->fetchCounters();
        if (c->h_cnt.errInvert) {                                  // not all of one length after all: count, then walk again
            HIP_CHECK(hipMemsetAsync(&c->d_cnt->errInvert, 0, sizeof(u64), c->stream));
            count();
            walk();
        }
    }
}
