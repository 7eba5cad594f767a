// k_dnac.hip -- the BFQDNAC1 container: read-order DNA (OUT.fq.dna, BFQzip.py:19-21,253-275: what the reference hands to
// 7z PPMd / bsc) through a block-adaptive hashed order-16 model + rANS.  oracle/bfq_codec_ref.c states the format; this
// file produces the same bytes.
//
// Why a second container: the redundancy of a 30x collection is between reads that cover the same stretch of the genome, 16
// and more symbols of context away from what the static order-7 table of BFQRANS2 sees (2.06 bits per base).  An adaptive
// model is sequential symbol by symbol; here it adapts block by block instead: a block of segments is coded against the
// table as it stood before the block (one lane per segment, the rows it needs fetched from a table of up to 2^30 rows in
// HBM), then the table takes the block in (a row is one 64-bit word of five 12-bit counters, an update one atomic add that
// nobody waits for: any order of the updates gives the same table).
// Nothing of the model is stored -- the decoder rebuilds it from what it has decoded.
//
//   k_dnac_lens      lengths of the lines (one lane per line)
//   k_dnac_syms      lines -> bases 0..4 without the newlines (one lane per line) / the way back (k_dnac_lines)
//   k_dnac_segfirst  first read of every 1024-base window (binary search in the read offsets)
//   k_dnac_encode    one lane per segment: forward over its reads (row -> frequencies, the pushed symbol), then rANS backwards
//   k_dnac_decode    one lane per segment: row -> frequencies -> symbol, forwards
//   k_dnac_update    one lane per read: both strands' counters
#include <vector>
#include <string.h>
#include "bfq_internal.h"
#include "bfq_device.h"

#define DQ_S 1024u
#define DQ_W 64u
#define DQ_KPLUS 1u
#define DQ_TSKIP 8u
#define DQ_AHEAD 8u                 // rows the encoder fetches ahead of the symbol it is at
#define DQ_HDR 72u
#define DQ_SCALE 12u
#define DQ_L (1u << 23)
#define DQ_MAXLINE 65535u
#define DQ_SLOT(cnt) (2ull * (cnt) + 16ull)          // scratch bytes of a segment of cnt symbols (a symbol costs at most 12 bits)

u64 bfq_rans_compress_device(bfq_ctx *c, const u8 *d_in, u64 n, u8 *d_out, u64 cap, bool dry);              // k_codec.hip
u64 bfq_rans_decompress_device(bfq_ctx *c, const u8 *h_in, const u8 *d_in, u64 len, u8 *d_out, u64 cap);
u64 bfq_codec_checksum_device(bfq_ctx *c, const u8 *d_in, u64 n, u64 *d_tmp);

static u32 dq_H(u64 nbases)
{
    u32 H = 12;
    while (H < 32 && (1ull << H) < nbases) H++;
    return H;
}
static u32 dq_K(u64 nbases)
{
    u32 l4 = 0;
    while (l4 < 32 && (1ull << (2 * l4)) < nbases) l4++;
    const u32 K = l4 + DQ_KPLUS;
    return K < 10 ? 10 : K > 20 ? 20 : K;
}
struct DqPar { u32 K, H, W, cap, tskip; u64 M; };
static DqPar dq_make(u32 K, u32 H, u32 W, u32 tskip) { return DqPar{K, H, W, 4080u / W, tskip ? tskip : 0xFFFFu, (1ull << (3 * K)) - 1ull}; }
static u64 dq_block_segs(u64 b, u64 nseg)
{
    u64 cap = nseg / 64;
    cap = cap < 256 ? 256 : cap > 65536 ? 65536 : cap;
    const u64 v = 16ull << (b < 12 ? b : 12);
    return v > cap ? cap : v;
}
static void put32(u8 *p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); p[3] = (u8)(v >> 24); }
static void put64(u8 *p, u64 v) { put32(p, (u32)v); put32(p + 4, (u32)(v >> 32)); }
static u32 get32(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
static u64 get64(const u8 *p) { return (u64)get32(p) | ((u64)get32(p + 4) << 32); }

__device__ __forceinline__ u64 dq_mix64(u64 z) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z; }
__device__ __forceinline__ u64 dq_row(u64 ctx, u32 kk, u32 H) { return dq_mix64(ctx * 32 + kk) >> (64 - H); }
// floor(x / T) for x < 2^24, 0 < T < 2^16: one float reciprocal and a correction (exact)
__device__ __forceinline__ u32 dq_div(u32 x, u32 T, float rT)
{
    u32 q = (u32)((float)x * rT);
    int r = (int)x - (int)(q * T);
    if (r < 0) { q--; r += (int)T; }
    if (r >= (int)T) q++;
    return q;
}
// the five frequencies of a row (they sum to 2^12) and the row's error verdict on symbol `c` at position j
__device__ __forceinline__ u32 dq_count(u64 row, int s, const DqPar &P) { const u32 v = (u32)(row >> (12 * s)) & 0xFFFu; return v > P.cap ? P.cap : v; }
__device__ __forceinline__ void dq_freqs(u64 row, const DqPar &P, u32 *f)
{
    u32 v[5], T = 0;
#pragma unroll
    for (int s = 0; s < 5; s++) { v[s] = dq_count(row, s, P) * P.W + (s < 4 ? 3u : 1u); T += v[s]; }
    const float rT = 1.0f / (float)T;
    u32 sum = 0, best = 0, bv = 0;
#pragma unroll
    for (int s = 0; s < 5; s++) { f[s] = dq_div(v[s] * ((1u << DQ_SCALE) - 5u), T, rT) + 1u; sum += f[s]; if (f[s] > bv) { bv = f[s]; best = (u32)s; } }
#pragma unroll
    for (int s = 0; s < 5; s++) if ((u32)s == best) f[s] += (1u << DQ_SCALE) - sum;   // the largest one (lowest symbol among equals)
}
__device__ __forceinline__ u32 dq_push(u64 row, const DqPar &P, u32 j, u32 c)
{
    if (j < P.K) return c;
    u32 m = 0, cm = 0, cc = 0, tot = 0;
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const u32 v = dq_count(row, s, P);
        tot += v;
        if (s < 4 && (s == 0 || v > cm)) { cm = v; m = (u32)s; }
        if ((u32)s == c) cc = v;
    }
    return (cm >= 3u && cc == 0u && tot - cm <= cm / 8u) ? m : c;
}
// bit 3 of a pushed symbol: the frozen row knows the base well already (count >= tskip) -- the table is not updated for it
__device__ __forceinline__ u32 dq_known(u64 row, const DqPar &P, u32 c) { return dq_count(row, (int)c, P) >= P.tskip ? 8u : 0u; }

__global__ __launch_bounds__(256) void k_dnac_lens(const u64 *__restrict__ lineEnd, u64 nreads, u32 *__restrict__ lens, u32 *__restrict__ bad)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nreads; i += (u64)gridDim.x * blockDim.x) {
        const u64 s = i ? lineEnd[i - 1] + 1 : 0, len = lineEnd[i] - s;
        if (len > DQ_MAXLINE) atomicOr(bad, 1u);
        lens[i] = (u32)len;
    }
}
// lines -> symbols; any byte that is no base sets the flag
__global__ __launch_bounds__(256) void k_dnac_syms(const u8 *__restrict__ in, const u64 *__restrict__ boff, u64 nreads, u8 *__restrict__ sym,
                                                   u32 *__restrict__ bad)
{
    bool b = false;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nreads; i += (u64)gridDim.x * blockDim.x) {
        const u64 o = boff[i], len = boff[i + 1] - o;
        const u8 *src = in + o + i;
        for (u64 j = 0; j < len; j++) {
            const u32 ch = src[j];
            const u32 s = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : ch == 'N' ? 4u : 7u;
            if (s == 7u) b = true;
            sym[o + j] = (u8)(s & 4u ? 4u : s);
        }
    }
    if (b) atomicOr(bad, 1u);
}
__global__ __launch_bounds__(256) void k_dnac_lines(const u8 *__restrict__ sym, const u64 *__restrict__ boff, u64 nreads, u8 *__restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nreads; i += (u64)gridDim.x * blockDim.x) {
        const u64 o = boff[i], len = boff[i + 1] - o;
        u8 *dst = out + o + i;
        for (u64 j = 0; j < len; j++) { const u32 s = sym[o + j]; dst[j] = (u8)(s == 0 ? 'A' : s == 1 ? 'C' : s == 2 ? 'G' : s == 3 ? 'T' : 'N'); }
        dst[len] = '\n';
    }
}
// segFirst[g], g = 0..nseg: the first read whose first base has an index >= g S (nreads when there is none)
__global__ __launch_bounds__(256) void k_dnac_segfirst(const u64 *__restrict__ boff, u64 nreads, u64 nseg, u64 *__restrict__ segFirst)
{
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g <= nseg; g += (u64)gridDim.x * blockDim.x) {
        u64 lo = 0, hi = nreads;
        const u64 want = g * DQ_S;
        while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (boff[mid] >= want) hi = mid; else lo = mid + 1; }
        segFirst[g] = lo;
    }
}

struct DqBlock {
    const u64 *T;           // the table: 2^H rows of 8 bytes
    DqPar P;
    const u64 *boff;        // [nreads + 1]
    const u64 *segFirst;    // [nseg + 1]
    u64 g0, g1;             // segments of this block
    u64 base0;              // first base of the block
    u8 *sym, *psh;          // [nbases]
};

// One lane per segment.  Forward: frequency | cumulative << 16 of every symbol into the lane's part of fc[], the pushed symbols.
// Backward: rANS from the last symbol to the first, bytes collected eight at a time towards lower addresses of the lane's slot.
__global__ __launch_bounds__(256) void k_dnac_encode(DqBlock B, u32 *__restrict__ fc, u8 *__restrict__ scratch, u32 *__restrict__ segBytes)
{
    for (u64 g = B.g0 + (u64)blockIdx.x * blockDim.x + threadIdx.x; g < B.g1; g += (u64)gridDim.x * blockDim.x) {
        const u64 ra = B.segFirst[g], rb = B.segFirst[g + 1];
        const u64 b0 = B.boff[ra], cnt = B.boff[rb] - b0;
        if (!cnt) { segBytes[g] = 0; continue; }
        u32 *myfc = fc + (b0 - B.base0);
        for (u64 r = ra; r < rb; r++) {
            const u64 rs = B.boff[r], len = B.boff[r + 1] - rs;
            u64 ctx = 0;
            // A context depends on the pushed symbols before it, and those on the rows before: a chain of HBM latencies.  But a
            // pushed symbol is the symbol itself 99 times in 100, so the next DQ_AHEAD rows are fetched at once for the contexts
            // the raw symbols give, and the group is cut short where a base was replaced (the rows behind it were the wrong ones).
            for (u64 j = 0; j < len;) {
                const u32 n = len - j < DQ_AHEAD ? (u32)(len - j) : DQ_AHEAD;
                u64 w = 0;                                         // the group's symbols, 8 bits each
                for (u32 t = 0; t < n; t++) w |= (u64)B.sym[rs + j + t] << (8 * t);
                u64 rows[DQ_AHEAD];
                {
                    u64 cs = ctx;
#pragma unroll
                    for (u32 t = 0; t < DQ_AHEAD; t++) {
                        const u64 jj = j + t;
                        const u32 kk = jj < B.P.K ? (u32)jj : B.P.K;
                        rows[t] = t < n ? B.T[dq_row(cs, kk, B.P.H)] : 0ull;
                        cs = ((cs << 3) | ((w >> (8 * t)) & 7u)) & B.P.M;
                    }
                }
                u32 done = 0;
#pragma unroll
                for (u32 t = 0; t < DQ_AHEAD; t++) {
                    if (t < n && done == t) {
                        const u64 row = rows[t];
                        u32 f[5];
                        dq_freqs(row, B.P, f);
                        const u32 s = (u32)(w >> (8 * t)) & 7u;
                        u32 cum = 0, fs = f[0];
#pragma unroll
                        for (int u = 1; u < 5; u++) { const bool past = (u32)u <= s; cum = past ? cum + f[u - 1] : cum; fs = (u32)u == s ? f[u] : fs; }
                        myfc[rs + j + t - b0] = fs | (cum << 16);
                        const u32 p = dq_push(row, B.P, (u32)(j + t), s);
                        B.psh[rs + j + t] = (u8)(p | dq_known(row, B.P, s));
                        ctx = ((ctx << 3) | p) & B.P.M;
                        done = p == s ? t + 1 : DQ_AHEAD + 1 + t;  // replaced: the rest of the group starts over
                    }
                }
                j += done > DQ_AHEAD ? done - DQ_AHEAD : done;
            }
        }
        u8 *const slotEnd = scratch + DQ_SLOT(b0 - B.base0 + cnt) + 16ull * (g - B.g0);      // slots in segment order: 2 bytes per symbol + 16 per segment
        u8 *q = slotEnd;
        u64 acc = 0;
        u32 nacc = 0, x = DQ_L;
        for (u64 i = cnt; i-- > 0;) {
            const u32 t = myfc[i];
            const u32 f = t & 0xFFFFu, c0 = t >> 16;
            const u32 xmax = ((DQ_L >> DQ_SCALE) << 8) * f;
            while (x >= xmax) {
                acc = (acc << 8) | (x & 0xFFu); x >>= 8;
                if (++nacc == 8) { q -= 8; __builtin_memcpy(q, &acc, 8); nacc = 0; }
            }
            const u32 d = x / f;
            x = (d << DQ_SCALE) + (x - d * f) + c0;
        }
        while (nacc) { nacc--; *--q = (u8)(acc >> (8u * nacc)); }
        q -= 4;
        q[0] = (u8)x; q[1] = (u8)(x >> 8); q[2] = (u8)(x >> 16); q[3] = (u8)(x >> 24);
        segBytes[g] = (u32)(slotEnd - q);
    }
}
// one wavefront per segment: its stream from the end of its scratch slot to its place in the payload (off[] is relative to
// the block, *runBase the payload bytes of the blocks before); bytes beyond `cap` are not written (the host sees the total)
__global__ __launch_bounds__(256) void k_dnac_pack(DqBlock B, const u8 *__restrict__ scratch, const u32 *__restrict__ segBytes,
                                                   const u64 *__restrict__ off, const u64 *__restrict__ runBase, u8 *__restrict__ out, u64 cap)
{
    const u32 lane = bfq_lane();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 g = B.g0 + (((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6); g < B.g1; g += nwaves) {
        const u32 bytes = segBytes[g];
        if (!bytes) continue;
        const u64 rb = B.segFirst[g + 1];
        const u8 *src = scratch + DQ_SLOT(B.boff[rb] - B.base0) + 16ull * (g - B.g0) - bytes;
        const u64 o = *runBase + off[g - B.g0];
        for (u32 j = lane; j < bytes; j += 64) if (o + j < cap) out[o + j] = src[j];
    }
}
__global__ void k_dnac_advance(u64 *runBase, const u64 *blockTotal) { *runBase += *blockTotal; }

// One lane per segment, forwards: the row of the context gives the frequencies, the state's low 12 bits the symbol.
__global__ __launch_bounds__(256) void k_dnac_decode(DqBlock B, const u8 *__restrict__ pay, const u64 *__restrict__ off,
                                                     const u32 *__restrict__ segBytes, u32 *__restrict__ bad)
{
    for (u64 g = B.g0 + (u64)blockIdx.x * blockDim.x + threadIdx.x; g < B.g1; g += (u64)gridDim.x * blockDim.x) {
        const u64 ra = B.segFirst[g], rb = B.segFirst[g + 1];
        const u32 nbytes = segBytes[g];
        if (B.boff[rb] == B.boff[ra]) { if (nbytes) atomicAdd(bad, 1u); continue; }
        if (nbytes < 4) { atomicAdd(bad, 1u); continue; }
        const u8 *q = pay + off[g];
        u32 x = (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16) | ((u32)q[3] << 24);
        if (x < DQ_L) { atomicAdd(bad, 1u); continue; }
        u32 used = 4, ni = 0;
        u64 ib = 0;
        bool ok = true;
        for (u64 r = ra; r < rb && ok; r++) {
            const u64 rs = B.boff[r], len = B.boff[r + 1] - rs;
            u64 ctx = 0;
            for (u64 j = 0; j < len; j++) {
                const u32 kk = j < B.P.K ? (u32)j : B.P.K;
                const u64 row = B.T[dq_row(ctx, kk, B.P.H)];
                u32 f[5];
                dq_freqs(row, B.P, f);
                const u32 slot = x & ((1u << DQ_SCALE) - 1u);
                u32 s = 0, c0 = 0, fs = f[0];
#pragma unroll
                for (int t = 1; t < 5; t++) { const bool past = slot >= c0 + fs; c0 = past ? c0 + fs : c0; s = past ? (u32)t : s; fs = past ? f[t] : fs; }
                x = fs * (x >> DQ_SCALE) + slot - c0;
                while (x < DQ_L) {
                    if (used >= nbytes + 4) { ok = false; break; }  // a damaged stream reads a few zeros past its segment, then stops
                    if (ni == 0) {
                        ib = 0;
                        if (used + 8 <= nbytes) __builtin_memcpy(&ib, q + used, 8);
                        else for (u32 t = 0; used + t < nbytes; t++) ib |= (u64)q[used + t] << (8 * t);
                        ni = 8;
                    }
                    x = (x << 8) | (u32)(ib & 0xFFu); ib >>= 8; ni--; used++;
                }
                if (!ok) break;
                B.sym[rs + j] = (u8)s;
                const u32 p = dq_push(row, B.P, (u32)j, s);
                B.psh[rs + j] = (u8)(p | dq_known(row, B.P, s));
                ctx = ((ctx << 3) | p) & B.P.M;
            }
        }
        if (used > nbytes) ok = false;
        if (!ok) atomicAdd(bad, 1u);
    }
}

// the table takes the reads [r0, r1) in: one lane per read, both strands; an update is one 64-bit atomic add without a return
// value (fields that pass 4095 carry into their neighbour: the format says so)
__global__ __launch_bounds__(256) void k_dnac_update(u64 *__restrict__ T, DqPar P, const u64 *__restrict__ boff, u64 r0, u64 r1,
                                                     const u8 *__restrict__ sym, const u8 *__restrict__ psh)
{
    for (u64 r = r0 + (u64)blockIdx.x * blockDim.x + threadIdx.x; r < r1; r += (u64)gridDim.x * blockDim.x) {
        const u64 rs = boff[r], len = boff[r + 1] - rs;
        u64 ctx = 0, rc = 0;          // rc = sum (3 - p[j - t]) << 3 (K - 1 - t): the newest pushed symbol on top
        u32 valid = 0;                // pushed symbols since the last one that is no base
        for (u64 j = 0; j < len; j++) {
            const u32 kk = j < P.K ? (u32)j : P.K;
            const u32 pb = psh[rs + j], p = pb & 7u;
            const bool skip = (pb & 8u) != 0;
            if (!skip) __hip_atomic_fetch_add(T + dq_row(ctx, kk, P.H), 1ull << (12u * sym[rs + j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctx = ((ctx << 3) | p) & P.M;
            rc = (rc >> 3) | ((u64)(3u - (p & 3u)) << (3 * (P.K - 1)));
            valid = p < 4u ? valid + 1u : 0u;
            if (!skip && j >= P.K && valid > P.K)
                __hip_atomic_fetch_add(T + dq_row(rc, P.K, P.H), 1ull << (12u * (3u - (psh[rs + j - P.K] & 7u))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
// arena when it has room, an allocation of its own otherwise (the fused job runs the codec in what the pipeline left)
struct DqMem {
    bfq_ctx *c;
    std::vector<void *> own;
    explicit DqMem(bfq_ctx *c) : c(c) {}
    ~DqMem() { for (void *p : own) (void)hipFree(p); }
    template <typename Tp> Tp *get(u64 count)
    {
        const u64 bytes = count * sizeof(Tp) + 256;
        if (c->wsCap - c->wsTop >= bytes + 512) return c->alloc<Tp>(count);
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); throw BfqError{BFQ_E_NOMEM, "stream codec: no device memory for the context table"}; }
        own.push_back(p);
        return (Tp *)p;
    }
};

// does the BFQDNAC1 container apply to d_in?  (lines of A C G T N; 64 KiB and more; 15 bases or more per line on average, none
// beyond 65535).  When it does: *nreads, and lens / boff / sym are built (from M).
struct DqInput { u64 nreads = 0, nbases = 0; u32 *lens = nullptr; u64 *boff = nullptr; u8 *sym = nullptr; };
static bool dq_prepare(bfq_ctx *c, DqMem &M, const u8 *d_in, u64 n, DqInput &I)
{
    if (n < 65536 || c->env.dnaStatic) return false;
    // under a workspace cap the context table (8 bytes per base, up to 34 GB) must fit what the arena has left: else the static container
    if (c->wsLimit() && c->wsCap - c->wsTop < (8ull << dq_H(n)) + 4 * n + (256u << 20)) return false;
    u8 last = 0;
    HIP_CHECK(hipMemcpyAsync(&last, d_in + n - 1, 1, hipMemcpyDeviceToHost, c->stream));
    const u64 nl = bfq_fastq_count_lines(c, d_in, n) - 1;        // (synchronises)
    if (last != 10 || nl == 0 || nl * 16 > n) return false;
    u64 nl2 = 0;
    const u64 *lineEnd = bfq_line_index(c, d_in, n, &nl2);
    I.nreads = nl; I.nbases = n - nl;
    I.lens = M.get<u32>(nl + 1);
    I.boff = M.get<u64>(nl + 2);
    I.sym = M.get<u8>(I.nbases + 16);
    u32 *d_bad = M.get<u32>(1);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, c->stream));
    KLAUNCH(c, K_CODEC, 8.0 * (double)nl, k_dnac_lens, bfq_grid(nl, 256), 256, lineEnd, nl, I.lens, d_bad);
    bfq_exscan_u32(c, I.lens, I.boff, nl, I.boff + nl);
    KLAUNCH(c, K_CODEC, 2.0 * (double)n, k_dnac_syms, bfq_grid(nl, 256), 256, d_in, (const u64 *)I.boff, nl, I.sym, d_bad);
    u32 bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    return bad == 0;
}

// first base of the segments g0 (one entry per block boundary): boff[segFirst[g]]
__global__ void k_dnac_bounds(const u64 *__restrict__ boff, const u64 *__restrict__ segFirst, const u64 *__restrict__ gs, u32 nb, u64 *__restrict__ out)
{
    for (u32 i = threadIdx.x; i < nb; i += blockDim.x) { out[2 * i] = segFirst[gs[i]]; out[2 * i + 1] = boff[segFirst[gs[i]]]; }
}
struct DqPlan { std::vector<u64> g, firstRead, firstBase; u64 maxBases = 0, maxSegs = 0; };
static void dq_plan(bfq_ctx *c, DqMem &M, const u64 *boff, const u64 *segFirst, u64 nseg, DqPlan &P)
{
    P.g.push_back(0);
    for (u64 b = 0; P.g.back() < nseg; b++) { const u64 e = P.g.back() + dq_block_segs(b, nseg); P.g.push_back(e < nseg ? e : nseg); }
    const u32 nb = (u32)P.g.size();
    u64 *d_g = M.get<u64>(nb), *d_o = M.get<u64>(2ull * nb);
    HIP_CHECK(hipMemcpyAsync(d_g, P.g.data(), 8ull * nb, hipMemcpyHostToDevice, c->stream));
    k_dnac_bounds<<<1, 256, 0, c->stream>>>(boff, segFirst, d_g, nb, d_o);
    std::vector<u64> o(2ull * nb);
    HIP_CHECK(hipMemcpyAsync(o.data(), d_o, 16ull * nb, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    P.firstRead.resize(nb); P.firstBase.resize(nb);
    for (u32 i = 0; i < nb; i++) { P.firstRead[i] = o[2 * i]; P.firstBase[i] = o[2 * i + 1]; }
    for (u32 i = 0; i + 1 < nb; i++) {
        if (P.firstBase[i + 1] - P.firstBase[i] > P.maxBases) P.maxBases = P.firstBase[i + 1] - P.firstBase[i];
        if (P.g[i + 1] - P.g[i] > P.maxSegs) P.maxSegs = P.g[i + 1] - P.g[i];
    }
}

// device workspace the container needs beyond what bfq_codec_workspace() counts for BFQRANS2 (the table)
u64 bfq_dnac_workspace(u64 n) { return n >= 65536 ? (8ull << dq_H(n)) + (64u << 20) : 0; }
static DqPar dq_default_par(bfq_ctx *c, u64 nbases)
{
    u32 K = dq_K(nbases), H = dq_H(nbases), W = DQ_W, ts = DQ_TSKIP;
    if (c->env.dnacK >= 8 && c->env.dnacK <= 20) K = (u32)c->env.dnacK;                     // (experiments; the header carries all four)
    if (c->env.dnacH >= 12 && c->env.dnacH <= 32) H = (u32)c->env.dnacH;
    if (c->env.dnacW >= 1 && c->env.dnacW <= 64) W = (u32)c->env.dnacW;
    if (c->env.dnacSkip >= 0 && c->env.dnacSkip <= 255) ts = (u32)c->env.dnacSkip;
    return dq_make(K, H, W, ts);
}

// d_in: n raw bytes on the device.  Returns the container's length, 0 when the container does not apply to the stream.
u64 bfq_dnac_compress_device(bfq_ctx *c, const u8 *d_in, u64 n, u8 *d_out, u64 cap, u64 *nbasesOut)
{
    const size_t mk = c->mark();
    DqMem M(c);
    DqInput I;
    if (!dq_prepare(c, M, d_in, n, I)) { c->release(mk); return 0; }
    *nbasesOut = I.nbases;
    const BfqError small{BFQ_E_ARG, "output buffer too small for the compressed stream"};
    const u64 nseg = (I.nbases + DQ_S - 1) / DQ_S;
    if (nseg > 0xFFFFFFFFull || cap < DQ_HDR) throw small;
    const DqPar Pm = dq_default_par(c, I.nbases);
    const u32 H = Pm.H;
    const u64 checksum = bfq_codec_checksum_device(c, d_in, n, M.get<u64>(1));
    const u64 ll = bfq_rans_compress_device(c, (const u8 *)I.lens, 4 * I.nreads, d_out + DQ_HDR, cap - DQ_HDR, false);
    const u64 hdr = DQ_HDR + ll + 4 * nseg;
    if (hdr > cap) throw small;
    u64 *T = M.get<u64>(1ull << H);
    HIP_CHECK(hipMemsetAsync(T, 0, 8ull << H, c->stream));
    u8 *psh = M.get<u8>(I.nbases + 16);
    u64 *segFirst = M.get<u64>(nseg + 2);
    u32 *segBytes = M.get<u32>(nseg + 1);
    KLAUNCH(c, K_CODEC, 8.0 * (double)nseg, k_dnac_segfirst, bfq_grid(nseg + 1, 256), 256, (const u64 *)I.boff, I.nreads, nseg, segFirst);
    DqPlan P;
    dq_plan(c, M, I.boff, segFirst, nseg, P);
    u32 *fc = M.get<u32>(P.maxBases + 16);
    u8 *scratch = M.get<u8>(DQ_SLOT(P.maxBases) + 16 * P.maxSegs + 64);
    u64 *off = M.get<u64>(P.maxSegs + 1), *d_tot = M.get<u64>(2);
    HIP_CHECK(hipMemsetAsync(d_tot, 0, 16, c->stream));
    u64 *runBase = d_tot + 1;
    for (size_t b = 0; b + 1 < P.g.size(); b++) {
        DqBlock B{T, Pm, I.boff, segFirst, P.g[b], P.g[b + 1], P.firstBase[b], I.sym, psh};
        const u64 ns = B.g1 - B.g0, nbz = P.firstBase[b + 1] - P.firstBase[b];
        if (!ns) continue;
        KLAUNCH(c, K_CODEC, 72.0 * (double)nbz, k_dnac_encode, bfq_grid(ns, 64), 64, B, fc, scratch, segBytes);
        bfq_exscan_u32(c, segBytes + B.g0, off, ns, d_tot);
        KLAUNCH(c, K_CODEC, 0.0, k_dnac_pack, bfq_grid(ns * 64, 256), 256, B, (const u8 *)scratch, (const u32 *)segBytes, (const u64 *)off,
                (const u64 *)runBase, d_out + hdr, cap - hdr);
        k_dnac_advance<<<1, 1, 0, c->stream>>>(runBase, d_tot);
        if (P.firstRead[b + 1] > P.firstRead[b])
            KLAUNCH(c, K_CODEC, 128.0 * 2.0 * (double)nbz, k_dnac_update, bfq_grid(P.firstRead[b + 1] - P.firstRead[b], 256), 256, T, Pm,
                    (const u64 *)I.boff, P.firstRead[b], P.firstRead[b + 1], (const u8 *)I.sym, (const u8 *)psh);
    }
    u64 total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, runBase, 8, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (hdr + total > cap) throw small;
    u8 h[DQ_HDR];
    memcpy(h, "BFQDNAC1", 8); put64(h + 8, n); put64(h + 16, I.nreads); put64(h + 24, I.nbases);
    put32(h + 32, Pm.K); put32(h + 36, H); put32(h + 40, DQ_S); put32(h + 44, (u32)nseg); put32(h + 48, DQ_SCALE); put32(h + 52, Pm.W | ((Pm.tskip == 0xFFFFu ? 0u : Pm.tskip) << 8));
    put64(h + 56, checksum); put64(h + 64, ll);
    HIP_CHECK(hipMemcpyAsync(d_out, h, DQ_HDR, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(d_out + DQ_HDR + ll, segBytes, 4 * nseg, hipMemcpyDeviceToDevice, c->stream));
    c->sync();
    c->release(mk);
    return hdr + total;
}

u64 bfq_dnac_member_len(const u8 *h_in, u64 len)
{
    const BfqError bad{BFQ_E_ARG, "not a BFQDNAC1 stream (or a damaged one)"};
    if (len < DQ_HDR || memcmp(h_in, "BFQDNAC1", 8)) throw bad;
    const u64 nseg = get32(h_in + 44), ll = get64(h_in + 64);
    if (ll > len || DQ_HDR + ll + 4 * nseg > len) throw bad;
    u64 total = DQ_HDR + ll + 4 * nseg;
    for (u64 g = 0; g < nseg; g++) total += get32(h_in + DQ_HDR + ll + 4 * g);
    if (total > len) throw bad;
    return total;
}

// h_in: the whole container on the host, d_in: the same bytes on the device.  The raw bytes go to d_out; returns their number.
u64 bfq_dnac_decompress_device(bfq_ctx *c, const u8 *h_in, const u8 *d_in, u64 len, u8 *d_out, u64 cap)
{
    const BfqError bad{BFQ_E_ARG, "damaged BFQDNAC1 stream"};
    (void)bfq_dnac_member_len(h_in, len);
    const u64 n = get64(h_in + 8), nreads = get64(h_in + 16), nbases = get64(h_in + 24), ll = get64(h_in + 64);
    const u32 K = get32(h_in + 32), H = get32(h_in + 36), nseg = get32(h_in + 44), W = get32(h_in + 52) & 0xFFu, ts = (get32(h_in + 52) >> 8) & 0xFFu;
    if (n > cap) throw BfqError{BFQ_E_ARG, "output buffer too small for the raw stream"};
    if (nreads == 0 || nreads > n || nbases != n - nreads || K < 8 || K > 20 || H < 12 || H > 32 || get32(h_in + 40) != DQ_S ||
        get32(h_in + 48) != DQ_SCALE || W < 1 || W > 64 || (get32(h_in + 52) >> 16) || nseg != (nbases + DQ_S - 1) / DQ_S)
        throw bad;
    const DqPar Pm = dq_make(K, H, W, ts);
    if (bfq_codec_raw_len(h_in + DQ_HDR, ll) != 4 * nreads || bfq_codec_member_len(h_in + DQ_HDR, ll) != ll) throw bad;
    const size_t mk = c->mark();
    DqMem M(c);
    u32 *lens = M.get<u32>(nreads + 4);
    u64 *boff = M.get<u64>(nreads + 2);
    if (bfq_rans_decompress_device(c, h_in + DQ_HDR, d_in + DQ_HDR, ll, (u8 *)lens, 4 * nreads) != 4 * nreads) throw bad;
    bfq_exscan_u32(c, lens, boff, nreads, boff + nreads);
    u64 tb = 0;
    HIP_CHECK(hipMemcpyAsync(&tb, boff + nreads, 8, hipMemcpyDeviceToHost, c->stream));
    u32 *segBytes = M.get<u32>(nseg + 1), *d_bad = M.get<u32>(1);
    u64 *off = M.get<u64>(nseg + 1), *segFirst = M.get<u64>((u64)nseg + 2);
    HIP_CHECK(hipMemcpyAsync(segBytes, h_in + DQ_HDR + ll, 4ull * nseg, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, c->stream));
    c->sync();
    if (tb != nbases) throw bad;
    if (nseg) bfq_exscan_u32(c, segBytes, off, nseg, nullptr);
    u8 *sym = M.get<u8>(nbases + 16), *psh = M.get<u8>(nbases + 16);
    u64 *T = M.get<u64>(1ull << H);
    HIP_CHECK(hipMemsetAsync(T, 0, 8ull << H, c->stream));
    KLAUNCH(c, K_CODEC, 8.0 * (double)nseg, k_dnac_segfirst, bfq_grid((u64)nseg + 1, 256), 256, (const u64 *)boff, nreads, (u64)nseg, segFirst);
    DqPlan P;
    dq_plan(c, M, boff, segFirst, nseg, P);
    const u8 *pay = d_in + DQ_HDR + ll + 4ull * nseg;
    for (size_t b = 0; b + 1 < P.g.size(); b++) {
        DqBlock B{T, Pm, boff, segFirst, P.g[b], P.g[b + 1], P.firstBase[b], sym, psh};
        const u64 ns = B.g1 - B.g0, nbz = P.firstBase[b + 1] - P.firstBase[b];
        if (!ns) continue;
        KLAUNCH(c, K_CODEC, 72.0 * (double)nbz, k_dnac_decode, bfq_grid(ns, 64), 64, B, pay, (const u64 *)off, (const u32 *)segBytes, d_bad);
        if (P.firstRead[b + 1] > P.firstRead[b])
            KLAUNCH(c, K_CODEC, 128.0 * 2.0 * (double)nbz, k_dnac_update, bfq_grid(P.firstRead[b + 1] - P.firstRead[b], 256), 256, T, Pm,
                    (const u64 *)boff, P.firstRead[b], P.firstRead[b + 1], (const u8 *)sym, (const u8 *)psh);
    }
    KLAUNCH(c, K_CODEC, 2.0 * (double)n, k_dnac_lines, bfq_grid(nreads, 256), 256, (const u8 *)sym, (const u64 *)boff, nreads, d_out);
    u32 nbad = 0;
    HIP_CHECK(hipMemcpyAsync(&nbad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    const u64 sum = nbad ? 0 : bfq_codec_checksum_device(c, d_out, n, M.get<u64>(1));
    c->release(mk);
    if (nbad) throw bad;
    if (sum != get64(h_in + 56)) throw BfqError{BFQ_E_ARG, "damaged BFQDNAC1 stream (checksum of the decoded bytes)"};
    return n;
}
