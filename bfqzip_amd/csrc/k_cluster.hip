// k_cluster.hip -- LCP flags, positional clusters, noise reduction and quality smoothing.
//
//   k_lcp_flags : in(r) = thr(r) && !min(r), the only thing the cluster scan of
//                 bfq_int.cpp:685-711 looks at; thr/min in the closed form that the
//                 suffix-tree navigation of bfq_int.cpp:139-181,183-300 produces and
//                 that bfq_ext.cpp:377-392 streams from the LCP file:
//                   thr[r] = r>=1 && LCP[r]>=K
//                   min[r] = 1<=r<=n-2 && LCP[r-1]>LCP[r] && LCP[r+1]>=LCP[r]
//   k_cluster   : every maximal run [b,e] of in() is the cluster [b-1,e]
//                 (process_cluster(begin,i), bfq_int.cpp:414-626, border=1); clusters
//                 are disjoint, so each one is handled independently by the thread
//                 that sits on its first in() row, walking it in row order exactly as
//                 the reference does (the M=1 double sum keeps its order).
//   k_big_*     : clusters longer than CL_BIG rows (low-complexity reads: one cluster can span
//                 millions of rows) are cut into tiles spread over the whole device: counts by
//                 atomics into a per-cluster state, one decision, edits tile by tile.  Same
//                 results; only the M=1 double sum stays sequential (its rounding depends on the order).
// Base replacements are recorded in modsym[r] (0 = untouched) instead of the
// reference's rankbv bit + BWT_MOD string (bfq_int.cpp:386-387,582-591).
#include <stdlib.h>
#include "bfq_internal.h"
#include "bfq_device.h"
#include "bfq_rank.h"
#include "bfq_rankblk.h"

// 8 rows per thread: ten LCP values in (two unaligned 8-byte loads + the two neighbours), eight flags out
__global__ __launch_bounds__(256) void k_lcp_flags(const u16 *__restrict__ lcp, u64 n, int K, u8 *__restrict__ in)
{
    const u64 ngroups = (n + 7) / 8;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (u64)gridDim.x * blockDim.x) {
        u64 r0 = g * 8;
        int l[10];                                   // l[k] = LCP[r0 - 1 + k]
        if (r0 >= 1 && r0 + 9 <= n) {
            u64 a = *(const u64 *)(lcp + r0 - 1), b = *(const u64 *)(lcp + r0 + 3);
            u32 c = *(const u32 *)(lcp + r0 + 7);
#pragma unroll
            for (int k = 0; k < 4; k++) { l[k] = (int)((a >> (16 * k)) & 0xFFFF); l[4 + k] = (int)((b >> (16 * k)) & 0xFFFF); }
            l[8] = (int)(c & 0xFFFF); l[9] = (int)(c >> 16);
        } else {
#pragma unroll
            for (int k = 0; k < 10; k++) { u64 r = r0 + k; l[k] = (r >= 1 && r - 1 < n) ? (int)lcp[r - 1] : 0; }
        }
        u64 out = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            u64 r = r0 + k;
            bool thr = (r >= 1) && (l[k + 1] >= K);
            bool mn = (r >= 1 && r + 2 <= n) && (l[k] > l[k + 1]) && (l[k + 2] >= l[k + 1]);
            out |= (u64)((thr && !mn) ? 1 : 0) << (8 * k);
        }
        if (r0 + 8 <= n) *(u64 *)(in + r0) = out;
        else for (u64 r = r0; r < n; r++) in[r] = (u8)(out >> (8 * (r - r0)));
    }
}

struct ClStat { u32 clust, disc, amb, mod, alleq, bases, qs, modb; };

struct BigState;
struct ClusterArgs {
    RankIndex R;
    const u8 *bwt; const u8 *qual; const u8 *in; u64 n;
    int m, v, f, t, term, M, ext;
    const double *powtab;   // [256] pow(10,-((signed char)q-33)/10), host libm
    const double *qthr;     // [qthrN] decreasing: smallest x with round(-10*log10(x)) <= qthrLo+k
    int qthrLo, qthrN;
    DevCounters *cnt;
    // position mode (w12 != nullptr): the rows still carry their sort records, so every row knows the text position of
    // its suffix; edits go straight to the output line streams at that position, the symbol before the preceding one
    // comes from the packed text instead of bwt[LF(row)].  No LF table is needed (R.lfq == nullptr).
    const u64 *w12;         // the records' (w1, w2) words in row order: position of the row's suffix in the terminated text
    const u64 *text3;       // packed text (bfq_common.h)
    u8 *outSym, *outQual;   // OUT.fq.dna / OUT.fq.qs layout = terminated text coordinates (read i at roff[i] + i, then '\n')
    int B;
    // rank mode (rankBlk != nullptr; neither an LF table nor sort records): a given eBWT under a workspace cap
    // (k_compact.hip).  LF(row) is answered on demand from the 64-byte rank blocks, smoothed qualities replace qual[] in
    // place and replaced bases go to repl[] (0 = untouched) -- the reference's own arrangement: QUAL[] edited in place,
    // rankbv bit + BWT_MOD (bfq_int.cpp:386-391).
    const u64 *rankBlk;
    u64 F[6];
    u8 *editQual, *repl;
    u64 *bigStart;          // first in() rows of the clusters left to the k_big_* kernels
    struct BigState *big;   // their state
    u16 *firstOut;          // per CL_WROWS rows: first row with in() == 0 (CL_WROWS: none), for k_big_extent
};

__device__ __forceinline__ int ord5(u8 c)   // bfq_int.cpp:106-110: A0 C1 G2 T3 N4
{
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
}
__device__ __forceinline__ u8 dna5(int i) { return (u8)((0x4E54474341ull >> (8 * i)) & 0xFF); }   // "ACGTN"

// Edits go to the LF-table entries the inversion reads (bfq_rank.h): the smoothed quality
// replaces the entry's quality byte, a replaced base sets the flag and the replacement
// code (reference: QUAL[j]=..., rankbv_setbit + BWT_MOD.push_back, bfq_int.cpp:386-391).
// a.qual[] keeps the original permuted qualities (read only).
__device__ __forceinline__ u64 row_pos(const ClusterArgs &a, u64 j)      // text position of row j's suffix
{
    const u64 x = a.w12[j];
    return bfq_val_pos(bfq_rec_pay((u32)x, (u32)(x >> 32)));
}
__device__ __forceinline__ void set_qual(const ClusterArgs &a, u64 j, int newqs)
{
    if (a.w12) { const u32 q = (u32)newqs & 0xFFu; a.outQual[row_pos(a, j) - 1] = (u8)(a.B ? bfq_bin8(q) : q); }
    else if (a.rankBlk) a.editQual[j] = (u8)newqs;
    else lfq_set_qual(a.R.lfq, j, (u32)newqs & 0xFFu);
}
__device__ __forceinline__ void set_mod(const ClusterArgs &a, u64 j, u8 sym)
{
    if (a.w12) a.outSym[row_pos(a, j) - 1] = sym;
    else if (a.rankBlk) a.repl[j] = sym;
    else lfq_set_repl(a.R.lfq, j, bfq_base_code(a.bwt[j]), bfq_base_code(sym));
}
// the symbol that precedes the eBWT symbol of row j: bwt[LF(j)] (bfq_int.cpp:545-560,577); row j holds a base
__device__ __forceinline__ u8 prec_sym(const ClusterArgs &a, u64 j)
{
    if (a.rankBlk) {                                           // LF(j) = F[c] + rank_c(j) from one rank block (dna_bwt_n.hpp:80-101)
        const RankBlk b = load_blk(a.rankBlk, j >> 6);
        const u32 code = blk_code(b, j);
        if (!code) return (u8)a.term;
        return a.bwt[a.F[code] + blk_occ_of(b, j, code)];
    }
    if (!a.w12) return a.bwt[lfq_next(a.R.lfq[j])];
    const u64 p = row_pos(a, j);                               // bwt[j] = text[p - 1], the one before it text[p - 2]
    if (p < 2) return (u8)a.term;
    const u64 t = p - 2, w = t / BFQ_SYMS_PER_WORD;
    const u32 code = (u32)(a.text3[bfq_t3_at(w)] >> (3u * (20u - (u32)(t - w * BFQ_SYMS_PER_WORD)))) & 7u;
    return code ? bfq_code_sym(code) : (u8)a.term;
}

// bfq_int.cpp:376-405 modBasesSmoothQS
__device__ __forceinline__ void mod_smooth(const ClusterArgs &a, u64 start, u64 end, u8 newSymb, int newqs, u32 lowQS, ClStat &st)
{
    const u8 TERM = (u8)a.term;
    for (u64 j0 = start; j0 <= end; j0 += 8) {                                 // 8 rows per (unaligned) 8-byte load
        u64 xb = *(const u64 *)(a.bwt + j0), xq = *(const u64 *)(a.qual + j0);
        u32 cnt = (end - j0 + 1 < 8) ? (u32)(end - j0 + 1) : 8u;
        for (u32 k = 0; k < cnt; k++) {
            u64 j = j0 + k;
            u8 b = (u8)(xb >> (8 * k));
            if (b == TERM) continue;
            if (b != newSymb && !((lowQS >> ord5(b)) & 1u)) { set_mod(a, j, newSymb); st.modb++; }
            else if (b == newSymb) { set_qual(a, j, newqs); st.qs++; }
            else if (newqs < (int)(signed char)(u8)(xq >> (8 * k))) { set_qual(a, j, newqs); st.qs++; }
        }
    }
}

// One cluster starting at row `start` (= first in() row - 1).  Its extent comes out of the same walk that
// counts it: row j > start belongs while in[start+1..j] are all set.  Returns true (nothing done) when the
// cluster outgrows CL_BIG rows: those go to the k_big_* kernels.
#define CL_BIG 2048
__device__ __forceinline__ bool process_cluster_body(const ClusterArgs &a, u64 start, ClStat &st)
{
    const u8 TERM = (u8)a.term;
    u32 freqs[5] = {0, 0, 0, 0, 0};
    u32 lowQS = 0;
    u64 base_num = 0;
    int sum = 0, mx = 0;                                                       // :323-338 avg_qs, :342-353 max_qs
    double sum_err = 0;                                                        // :357-373 mean_error (row order kept)
    u64 end;
    for (u64 j0 = start;; j0 += 8) {                                           // :437-449, 8 rows per 8-byte load
        u64 xin = *(const u64 *)(a.in + j0 + 1);                               // in() of rows j0+1 .. j0+8 (in[] is padded)
        u64 z = (xin - 0x0101010101010101ull) & ~xin & 0x8080808080808080ull;  // first zero byte
        u32 run = z ? (u32)(__builtin_ctzll(z) >> 3) : 8u;
        if (j0 + run >= a.n) run = (u32)(a.n - 1 - j0);
        u32 cnt = run < 8 ? run + 1 : 8u;                                      // rows j0 .. j0+cnt-1 are inside
        u64 xb = *(const u64 *)(a.bwt + j0), xq = *(const u64 *)(a.qual + j0);
        for (u32 k = 0; k < cnt; k++) {
            u8 b = (u8)(xb >> (8 * k));
            if (b != TERM) {
                int o = ord5(b);
                u8 qb = (u8)(xq >> (8 * k));
                int q = (int)(signed char)qb;
                freqs[o]++; base_num++;
                if (q >= a.t + 33) lowQS |= 1u << o;
                sum += q;
                if (q > mx) mx = q;
                if (a.M == 1) sum_err = sum_err + a.powtab[qb];
            }
        }
        if (run < 8) { end = j0 + run; break; }
        if (j0 + 8 - start >= CL_BIG) return true;
    }
    u64 size = end - start + 1;
    if (size < (u64)(long long)a.m) return false;                             // :422
    st.clust++;
    if (base_num == 0) return false;                                           // :453
    st.bases += (u32)base_num;

    int newqs;
    if (a.M == 1) {
        double avg_err = sum_err / (double)base_num;
        int lo = 0, hi = a.qthrN - 1;                                          // first k with qthr[k] <= avg_err
        while (lo < hi) { int mid = (lo + hi) >> 1; if (a.qthr[mid] <= avg_err) hi = mid; else lo = mid + 1; }
        int q = a.qthrLo + lo;
        newqs = a.ext ? (int)(signed char)(u8)((u8)q + 33) : (int)(signed char)(q + 33);
    } else if (a.M == 2) {
        newqs = (int)(signed char)a.v;                                         // :467
    } else if (a.M == 3) {
        if (sum == 0) newqs = 0;
        else if (a.ext) newqs = (int)(signed char)(u8)roundf((float)sum / (float)base_num);   // bfq_ext.cpp:496
        else newqs = (int)(signed char)(int)((u64)(long long)sum / base_num);
    } else {
        newqs = mx;
    }

    u8 Freq[5];
    int nf = 0, nnn = 0;                                                       // :480-499
    for (int s = 0; s < 5; s++)
        if (freqs[s] > 0) {
            nnn++;
            u32 perc = (u32)((100ull * freqs[s]) / base_num) & 0xFFu;
            if ((float)perc >= (float)a.f) Freq[nf++] = dna5(s);
        }
    if (nnn == 1) st.alleq++;
    if (nf >= 3) { atomicAdd(&a.cnt->errFreq3, 1ull); return false; }               // :505 assert

    if (nf == 0) { st.disc++; return false; }
    if (nf == 1) {
        if (Freq[0] == 'N') st.disc++;
        else mod_smooth(a, start, end, Freq[0], newqs, lowQS, st);
        return false;
    }
    if (base_num < (u64)(long long)a.m) { st.disc++; return false; }                   // :520
    if (Freq[0] == 'N') { mod_smooth(a, start, end, Freq[1], newqs, lowQS, st); st.mod++; return false; }
    if (Freq[1] == 'N') { mod_smooth(a, start, end, Freq[0], newqs, lowQS, st); st.mod++; return false; }

    // :542-565 the symbols preceding the two frequent bases: bwt[LF(j)]
    u8 symbPrec[2] = {0, 0};
    u32 fr[2] = {0, 0};
    for (u64 j = start; j <= end; j++) {
        u8 b = a.bwt[j];
        int w = (b == Freq[0]) ? 0 : ((b == Freq[1]) ? 1 : -1);
        if (w < 0) continue;
        u8 ch = prec_sym(a, j);
        if (ch != TERM && ch != 'N') { fr[w] |= 1u << ord5(ch); symbPrec[w] = ch; }
    }
    if (__popc(fr[0] & 15u) == 1 && __popc(fr[1] & 15u) == 1 && symbPrec[0] != symbPrec[1]) {   // :568
        st.mod++;
        for (u64 j = start; j <= end; j++) {
            u8 b = a.bwt[j];
            if (b == TERM) continue;
            if (b != Freq[0] && b != Freq[1] && !((lowQS >> ord5(b)) & 1u)) {
                u8 ch = prec_sym(a, j);
                if (ch == symbPrec[0]) { set_mod(a, j, Freq[0]); st.modb++; }
                else if (ch == symbPrec[1]) { set_mod(a, j, Freq[1]); st.modb++; }
            } else if (b == Freq[0] || b == Freq[1]) {
                set_qual(a, j, newqs); st.qs++;
            } else if (newqs < (int)(signed char)a.qual[j]) {
                set_qual(a, j, newqs); st.qs++;
            }
        }
    } else {
        st.amb++;
    }
    return false;
}

// ---- one workgroup per chunk of rows: the chunk's cluster starts (rows r with in(r) && !in(r-1))
// are compacted into LDS (per-wave ballots, one prefix over the four waves), then every thread
// takes clusters from that list (dense lanes) and walks them in row order exactly as the
// reference does.  A cluster belongs to the chunk it starts in; its rows may extend past the
// chunk end.  Statistics (bfq_int.cpp:53-62) stay in per-thread scalars until the kernel ends.
#define CL_CHUNK 4096
#define CL_WROWS (CL_CHUNK / 4)                       // rows per wave
__global__ __launch_bounds__(256) void k_cluster(ClusterArgs a, u64 nchunks)
{
    __shared__ u32 shst[8];
    __shared__ u16 starts[4][CL_WROWS];               // per wave: chunk-local start rows
    __shared__ u32 wn[4];
    const u32 lane = bfq_lane(), w = threadIdx.x >> 6;
    if (threadIdx.x < 8) shst[threadIdx.x] = 0;
    ClStat st = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u64 ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        // 16 rows per lane: one 16-byte load of in() + the flag before them; lane order = row order
        const u64 r0 = ch * CL_CHUNK + (u64)w * CL_WROWS + (u64)lane * (CL_WROWS / 64);
        u32 bits = 0;                                              // bit k: in(r0 + k), rows past n are 0
        if (r0 < a.n) {
            uint4 x = *(const uint4 *)(a.in + r0);                 // in[] is padded by 64 bytes
            u32 wds[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int b = 0; b < 4; b++) bits |= ((wds[q] >> (8 * b)) & 1u) << (4 * q + b);
            if (r0 + 16 > a.n) bits &= (1u << (u32)(a.n - r0)) - 1u;
        }
        {                                                          // first row of this wave's 1024 that is outside every cluster
            const u32 zb = ~bits & 0xFFFFu;
            const u64 any = __ballot(zb != 0);
            const int fl = any ? __builtin_ctzll(any) : -1;
            const u32 fz = (u32)__builtin_amdgcn_readlane((int)zb, fl < 0 ? 0 : fl);
            if (lane == 0) a.firstOut[ch * 4 + w] = (u16)(fl < 0 ? CL_WROWS : fl * (CL_WROWS / 64) + __builtin_ctz(fz));
        }
        u32 prev = (r0 >= 1 && r0 < a.n) ? (u32)a.in[r0 - 1] & 1u : 0u;
        u32 sb = bits & ~((bits << 1) | prev) & 0xFFFFu;           // cluster starts: in(r) && !in(r-1)
        u32 ns = (u32)__popc(sb);
        u32 incl = bfq_wave_incscan32(ns);
        u32 o = incl - ns;
        while (sb) { u32 k = (u32)__builtin_ctz(sb); starts[w][o++] = (u16)(w * CL_WROWS + lane * (CL_WROWS / 64) + k); sb &= sb - 1; }
        u32 cntw = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        if (lane == 0) wn[w] = cntw;
        __syncthreads();
        u32 n0 = wn[0], n1 = wn[1], n2 = wn[2], n3 = wn[3];
        u32 total = n0 + n1 + n2 + n3;
        for (u32 t = threadIdx.x; t < total; t += 256) {
            u32 ww = t < n0 ? 0u : (t < n0 + n1 ? 1u : (t < n0 + n1 + n2 ? 2u : 3u));
            u32 idx = t - (ww == 0 ? 0u : (ww == 1 ? n0 : (ww == 2 ? n0 + n1 : n0 + n1 + n2)));
            u64 r = ch * CL_CHUNK + starts[ww][idx];
            if (process_cluster_body(a, r - 1, st)) a.bigStart[atomicAdd(&a.cnt->bigClusters, 1ull)] = r;
        }
        __syncthreads();
    }
    if (st.clust) atomicAdd(&shst[0], st.clust);
    if (st.disc) atomicAdd(&shst[1], st.disc);
    if (st.amb) atomicAdd(&shst[2], st.amb);
    if (st.mod) atomicAdd(&shst[3], st.mod);
    if (st.alleq) atomicAdd(&shst[4], st.alleq);
    if (st.bases) atomicAdd(&shst[5], st.bases);
    if (st.qs) atomicAdd(&shst[6], st.qs);
    if (st.modb) atomicAdd(&shst[7], st.modb);
    __syncthreads();
    if (threadIdx.x < 8 && shst[threadIdx.x]) atomicAdd(&a.cnt->stats[threadIdx.x], (u64)shst[threadIdx.x]);
}

// ---- clusters longer than CL_BIG rows: tiles of CB_TILE rows spread over the whole device ------------
// Per cluster one BigState in global memory: extent, counters (atomics from the tiles), the decision.
//   k_big_extent : one workgroup per cluster finds its end (16 in() flags per thread and step)
//   k_big_count  : tiles -> symbol frequencies, trusted-quality mask, quality sum / max          (:437-449)
//   k_big_decide : one thread per cluster, the decision of :451-540 (M=1: the double sum, in row order)
//   k_big_prec   : clusters with two frequent bases: the symbols that precede them               (:542-565)
//   k_big_apply  : tiles -> edits                                                               (:376-405, :568-591)
// Tile t of cluster i goes to workgroup (t + i) mod grid, so that many medium clusters and one huge
// cluster both use every workgroup.
struct BigState {
    u64 start, end;           // rows [start, end]
    u64 freq[5];
    u32 low, sum, mx, fr0, fr1;
    int mode, newqs;          // mode 0 nothing, 1 mod_smooth towards sym0, 2 two frequent bases (undecided), 3 the same, confirmed
    u32 sym0, sym1;
};
#define CB_TILE 65536
#define CB_GRID 2048

// rows [ts, te] of one tile, 16 per thread and step: one 16-byte load each of the eBWT and of the qualities
// (both 16-byte aligned and padded past n)
#define CB_FOR_ROWS(...)                                                                               \
    for (u64 g0 = (ts & ~15ull) + (u64)threadIdx.x * 16; g0 <= te; g0 += 256 * 16) {                  \
        const uint4 xb4 = *(const uint4 *)(a.bwt + g0), xq4 = *(const uint4 *)(a.qual + g0);          \
        const u32 xbw[4] = {xb4.x, xb4.y, xb4.z, xb4.w}, xqw[4] = {xq4.x, xq4.y, xq4.z, xq4.w};        \
        _Pragma("unroll") for (int k = 0; k < 16; k++) {                                              \
            const u64 j = g0 + k;                                                                      \
            if (j < ts || j > te) continue;                                                            \
            const u8 b = (u8)(xbw[k >> 2] >> (8 * (k & 3)));                                           \
            const u8 qb = (u8)(xqw[k >> 2] >> (8 * (k & 3)));                                          \
            (void)qb;                                                                                  \
            __VA_ARGS__                                                                                       \
        }                                                                                              \
    }
// the tiles of all big clusters that fall to this workgroup; `body` sees BigState &S, tile rows [ts, te]
#define CB_FOR_TILES(...)                                                                              \
    {                                                                                                  \
        const u64 nbig = a.cnt->bigClusters;                                                           \
        for (u64 bi = 0; bi < nbig; bi++) {                                                            \
            BigState &S = a.big[bi];                                                                   \
            const u64 ntiles = (S.end - S.start) / CB_TILE + 1;                                        \
            for (u64 t = (blockIdx.x + gridDim.x - (u32)(bi % gridDim.x)) % gridDim.x; t < ntiles; t += gridDim.x) { \
                const u64 ts = S.start + t * CB_TILE;                                                  \
                const u64 te = (ts + CB_TILE - 1 < S.end) ? ts + CB_TILE - 1 : S.end;                  \
                __VA_ARGS__                                                                                   \
            }                                                                                          \
        }                                                                                              \
    }

__global__ __launch_bounds__(256) void k_big_extent(ClusterArgs a)
{
    __shared__ u64 shEnd;
    const u64 nbig = a.cnt->bigClusters;
    for (u64 bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
        const u64 r = a.bigStart[bi];                               // cluster = [r-1, e], e = last row of the in() run
        __syncthreads();
        if (threadIdx.x == 0) shEnd = ~0ull;
        __syncthreads();
        // the rest of r's 1024-row piece flag by flag, then whole pieces by their first-out entries
        {
            const u64 pend = ((r + 1) / CL_WROWS + 1) * CL_WROWS;
            for (u64 b0 = (r + 1) & ~15ull; b0 < pend; b0 += 256 * 16) {
                const u64 i0 = b0 + (u64)threadIdx.x * 16;
                if (i0 < pend && i0 < a.n + 16) {
                    const uint4 x = *(const uint4 *)(a.in + i0);
                    const u32 wds[4] = {x.x, x.y, x.z, x.w};
                    for (u32 k = 0; k < 16; k++) {
                        const u64 xr = i0 + k;
                        if (xr <= r) continue;
                        if (xr >= a.n || !((wds[k >> 2] >> (8 * (k & 3))) & 1u)) { atomicMin(&shEnd, xr); break; }
                    }
                }
            }
            __syncthreads();
            const u64 npieces = (a.n + CL_WROWS - 1) / CL_WROWS;
            for (u64 p0 = pend / CL_WROWS;; p0 += 256) {
                bool done = (shEnd != ~0ull);
                __syncthreads();
                if (done) break;                       // uniform
                const u64 pc = p0 + threadIdx.x;
                if (pc >= npieces) atomicMin(&shEnd, a.n);
                else { const u32 fo = a.firstOut[pc]; if (fo < CL_WROWS) atomicMin(&shEnd, pc * CL_WROWS + fo); }
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) {
            BigState z = {};
            z.start = r - 1; z.end = shEnd - 1;
            a.big[bi] = z;
        }
    }
}

__global__ __launch_bounds__(256) void k_big_count(ClusterArgs a)
{
    const u8 TERM = (u8)a.term;
    CB_FOR_TILES(
        u32 fA = 0, fC = 0, fG = 0, fT = 0, fN = 0, low = 0, sum = 0;
        int mx = 0;
        CB_FOR_ROWS(
            if (b != TERM) {
                const int q = (int)(signed char)qb;
                const int o = ord5(b);
                fA += (o == 0); fC += (o == 1); fG += (o == 2); fT += (o == 3); fN += (o == 4);
                if (q >= a.t + 33) low |= 1u << o;
                sum += (u32)q;                                                 // wraps like the reference's int
                if (q > mx) mx = q;
            })
        if (fA) atomicAdd(&S.freq[0], (u64)fA);
        if (fC) atomicAdd(&S.freq[1], (u64)fC);
        if (fG) atomicAdd(&S.freq[2], (u64)fG);
        if (fT) atomicAdd(&S.freq[3], (u64)fT);
        if (fN) atomicAdd(&S.freq[4], (u64)fN);
        if (low) atomicOr(&S.low, low);
        if (sum) atomicAdd(&S.sum, sum);
        if (mx) atomicMax(&S.mx, (u32)mx);)
}

__global__ __launch_bounds__(256) void k_big_decide(ClusterArgs a)
{
    const u8 TERM = (u8)a.term;
    const u64 nbig = a.cnt->bigClusters;
    for (u64 bi = (u64)blockIdx.x * blockDim.x + threadIdx.x; bi < nbig; bi += (u64)gridDim.x * blockDim.x) {
        BigState &S = a.big[bi];
        const u64 start = S.start, end = S.end;
        const u64 size = end - start + 1;
        const u64 base_num = S.freq[0] + S.freq[1] + S.freq[2] + S.freq[3] + S.freq[4];
        if (size < (u64)(long long)a.m) continue;
        atomicAdd(&a.cnt->stats[0], 1ull);
        if (!base_num) continue;
        atomicAdd(&a.cnt->stats[5], base_num);
        int newqs;
        if (a.M == 1) {                                                        // rounding depends on the order: one thread, row order
            double sum_err = 0;
            for (u64 j = start; j <= end; j++)
                if (a.bwt[j] != TERM) sum_err = sum_err + a.powtab[a.qual[j]];
            double avg_err = sum_err / (double)base_num;
            int lo = 0, hi = a.qthrN - 1;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (a.qthr[mid] <= avg_err) hi = mid; else lo = mid + 1; }
            int q = a.qthrLo + lo;
            newqs = a.ext ? (int)(signed char)(u8)((u8)q + 33) : (int)(signed char)(q + 33);
        } else if (a.M == 2) {
            newqs = (int)(signed char)a.v;
        } else if (a.M == 3) {
            int sum = (int)S.sum;
            if (sum == 0) newqs = 0;
            else if (a.ext) newqs = (int)(signed char)(u8)roundf((float)sum / (float)base_num);
            else newqs = (int)(signed char)(int)((u64)(long long)sum / base_num);
        } else {
            newqs = (int)S.mx;
        }
        S.newqs = newqs;
        u8 Freq[5];
        int nf = 0, nnn = 0;
        for (int s = 0; s < 5; s++)
            if (S.freq[s] > 0) {
                nnn++;
                u32 perc = (u32)((100ull * S.freq[s]) / base_num) & 0xFFu;
                if ((float)perc >= (float)a.f) Freq[nf++] = dna5(s);
            }
        if (nnn == 1) atomicAdd(&a.cnt->stats[4], 1ull);
        if (nf >= 3) atomicAdd(&a.cnt->errFreq3, 1ull);
        else if (nf == 0) atomicAdd(&a.cnt->stats[1], 1ull);
        else if (nf == 1) {
            if (Freq[0] == 'N') atomicAdd(&a.cnt->stats[1], 1ull);
            else { S.mode = 1; S.sym0 = Freq[0]; }
        } else if (base_num < (u64)(long long)a.m) atomicAdd(&a.cnt->stats[1], 1ull);
        else if (Freq[0] == 'N') { S.mode = 1; S.sym0 = Freq[1]; atomicAdd(&a.cnt->stats[3], 1ull); }
        else if (Freq[1] == 'N') { S.mode = 1; S.sym0 = Freq[0]; atomicAdd(&a.cnt->stats[3], 1ull); }
        else { S.mode = 2; S.sym0 = Freq[0]; S.sym1 = Freq[1]; }
    }
}

__global__ __launch_bounds__(256) void k_big_prec(ClusterArgs a)
{
    const u8 TERM = (u8)a.term;
    CB_FOR_TILES(
        if (S.mode != 2) continue;
        const u8 s0 = (u8)S.sym0; const u8 s1 = (u8)S.sym1;
        u32 f0 = 0, f1 = 0;
        CB_FOR_ROWS(
            if (b == s0 || b == s1) {
                const u8 ch = prec_sym(a, j);
                if (ch != TERM && ch != 'N') { if (b == s0) f0 |= 1u << ord5(ch); else f1 |= 1u << ord5(ch); }
            })
        if (f0) atomicOr(&S.fr0, f0);
        if (f1) atomicOr(&S.fr1, f1);)
}

__global__ __launch_bounds__(256) void k_big_apply(ClusterArgs a)
{
    const u8 TERM = (u8)a.term;
    u32 modb = 0, qs = 0;
    CB_FOR_TILES(
        const int mode = S.mode; const int newqs = S.newqs;
        const u32 lowQS = S.low;
        const u8 s0 = (u8)S.sym0; const u8 s1 = (u8)S.sym1;
        if (mode == 1) {                                                       // :376-405
            CB_FOR_ROWS(
                if (b != TERM) {
                    if (b != s0 && !((lowQS >> ord5(b)) & 1u)) { set_mod(a, j, s0); modb++; }
                    else if (b == s0) { set_qual(a, j, newqs); qs++; }
                    else if (newqs < (int)(signed char)qb) { set_qual(a, j, newqs); qs++; }
                })
        } else if (mode == 2) {                                                // :568-591
            const u32 f0 = S.fr0 & 15u; const u32 f1 = S.fr1 & 15u;
            // with exactly one preceding symbol per frequent base, "the last one seen" is that symbol
            const bool ok = __popc(f0) == 1 && __popc(f1) == 1 && f0 != f1;
            if (t == 0 && threadIdx.x == 0) atomicAdd(&a.cnt->stats[ok ? 3 : 2], 1ull);
            if (ok) {
                const u8 p0 = dna5(__ffs(f0) - 1); const u8 p1 = dna5(__ffs(f1) - 1);
                CB_FOR_ROWS(
                    if (b != TERM) {
                        if (b != s0 && b != s1 && !((lowQS >> ord5(b)) & 1u)) {
                            const u8 ch = prec_sym(a, j);
                            if (ch == p0) { set_mod(a, j, s0); modb++; }
                            else if (ch == p1) { set_mod(a, j, s1); modb++; }
                        } else if (b == s0 || b == s1) { set_qual(a, j, newqs); qs++; }
                        else if (newqs < (int)(signed char)qb) { set_qual(a, j, newqs); qs++; }
                    })
            }
        })
    if (qs) atomicAdd(&a.cnt->stats[6], (u64)qs);
    if (modb) atomicAdd(&a.cnt->stats[7], (u64)modb);
}

void bfq_lcp_flags(bfq_ctx *c, const u16 *lcp, u64 n, int K, u8 *in)
{
    if (!n) return;
    KLAUNCH(c, K_LCP_FLAGS, 3.0 * (double)n, k_lcp_flags, bfq_grid((n + 7) / 8, 256), 256, lcp, n, K, in);
}

void bfq_clusters(bfq_ctx *c, const RankIndex &R, const u8 *bwt, const u8 *qual, const u8 *in, u64 n, const ClusterPos *pm,
                  const ClusterRank *rm)
{
    if (!n) return;
    ClusterArgs a;
    a.R = R; a.bwt = bwt; a.qual = qual; a.in = in; a.n = n;
    a.rankBlk = rm ? rm->rankBlk : nullptr; a.editQual = rm ? rm->qual : nullptr; a.repl = rm ? rm->repl : nullptr;
    for (int s = 0; s < 6; s++) a.F[s] = rm ? rm->F[s] : 0;
    a.w12 = pm ? pm->w12 : nullptr; a.text3 = pm ? pm->text3 : nullptr;
    a.outSym = pm ? pm->outSym : nullptr; a.outQual = pm ? pm->outQual : nullptr; a.B = pm ? pm->B : 0;
    a.m = c->P.m; a.v = c->P.v; a.f = c->P.f; a.t = c->P.t; a.term = c->P.term & 0xFF; a.M = c->P.M; a.ext = c->P.ext;
    a.powtab = c->d_powtab; a.qthr = c->d_qthr; a.qthrLo = c->qthrLo; a.qthrN = c->qthrN;
    a.cnt = c->d_cnt;
    size_t mk = c->mark();
    // the list of long clusters is per call (pile by pile there are several calls between two resets of the counters)
    HIP_CHECK(hipMemsetAsync(&c->d_cnt->bigClusters, 0, sizeof(u64), c->stream));
    a.bigStart = c->alloc<u64>(n / CL_BIG + 2);
    a.big = c->alloc<BigState>(n / CL_BIG + 2);
    a.firstOut = c->alloc<u16>(n / CL_WROWS + 8);
    u64 nchunks = ceil_div(n, CL_CHUNK);
    KLAUNCH(c, K_CLUSTER, 4.125 * (double)n, k_cluster, bfq_grid(nchunks, 1), 256, a, nchunks);
    // the list length stays on the device: fixed grids stride over it (usually empty)
    KLAUNCH(c, K_CLUSTER_BIG, 0.0, k_big_extent, 1024, 256, a);
    KLAUNCH(c, K_CLUSTER_BIG, 0.0, k_big_count, CB_GRID, 256, a);
    KLAUNCH(c, K_CLUSTER_BIG, 0.0, k_big_decide, 64, 256, a);
    KLAUNCH(c, K_CLUSTER_BIG, 0.0, k_big_prec, CB_GRID, 256, a);
    KLAUNCH(c, K_CLUSTER_BIG, 0.0, k_big_apply, CB_GRID, 256, a);
    c->release(mk);
}
