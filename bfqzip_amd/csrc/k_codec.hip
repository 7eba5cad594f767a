// k_codec.hip -- the stream codec: entropy coding of OUT.fq.dna / OUT.fq.qs / OUT.h on the GPU, the step the
// reference hands to external tools (`7z a -mm=PPMd`, `bsc e ... -T`: step5 / step5b, BFQzip.py:253-275).
// SURVEY 8(f).4.  The container is this project's own (neither 7z nor libbsc sources are in the reference tree):
// a static order-k model of the whole stream + range-ANS, every segment of 8192 symbols on its own, so that
// encoder and decoder run one segment per lane.  oracle/bfq_codec_ref.c states the format and is what
// the tests compare this file with, byte for byte.
//
//   k_cdc_present : which byte values occur                                  (alphabet, dense symbol numbers)
//   k_cdc_count   : occurrences of every (context, symbol) pair in a SAMPLE of the segments (every S-th, S <= 64: a
//                   histogram of all symbols into millions of bins is atomics-bound, 47-170 ms per 0.9 G symbols, and
//                   the model does not need it -- every symbol keeps a non-zero share in every row and contexts
//                   the sample misses use the order-0 row); a lane walks its segment forwards and adds up
//                   runs of equal pairs before it touches the table (smoothed quality streams are long runs:
//                   one atomic per run, not per symbol)
//   host          : k, the model rows (normalised to 2^12), the cumulative rows -- the table is at most 4 M entries
//   k_cdc_encode  : a lane codes its segment last symbol to first (the context in front of every symbol follows
//                   from the one behind it without a second pass), bytes stored backwards into its scratch slot
//   k_cdc_pack    : the slots' streams closed up behind the header (one wavefront per segment)
//   k_cdc_decode  : a lane decodes its segment forwards
// Integer work, bound by the dependent table look-up per symbol; the input is read once per pass.
#include <string.h>
#include "bfq_internal.h"
#include "bfq_device.h"

#define CQ_SEG_MAX 8192u                             // symbols per segment; halved down to 1024 for short streams (cdc_choose_seg)
#define CQ_SCALE 12u
#define CQ_L (1u << 23)
#define CQ_SLOT(seg) (2u * (seg) + 16u)              // scratch bytes per segment (a symbol costs at most 12 bits)
#define CQ_MAX_TABLE (1u << 22)

__global__ __launch_bounds__(256) void k_cdc_present(const u8 *__restrict__ in, u64 n, u32 *__restrict__ present)
{
    __shared__ u32 sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) sh[in[i]] = 1;
    __syncthreads();
    if (sh[threadIdx.x]) present[threadIdx.x] = 1;
}

struct CdcModel {
    u32 A, k, nseg, seg;
    u32 top;                 // A^k
    u64 n;
};

// the k symbols in front of a position, newest in the low byte: the symbol that leaves the context is byte k-1
__device__ __forceinline__ u32 cdc_next_ctx(u32 ctx, u32 s, u64 &win, const CdcModel &m)
{
    if (!m.k) return 0;
    const u32 outgoing = (u32)(win >> (8u * (m.k - 1u))) & 0xFFu;
    win = (win << 8) | s;
    return ctx * m.A + s - outgoing * m.top;
}

// Hot (context, symbol) pairs -- a header stream has a few dozen, a smoothed quality stream a handful -- would serialise on
// their table entries (77 M atomics on such keys took 70 ms).  Every workgroup therefore owns a direct-mapped cache of 4096
// pairs in LDS: the first pair to claim a slot counts there (LDS atomics), everything else goes to the table; the caches
// are added to the table when the workgroup is done.  The sums are exact either way.
#define CQ_CACHE 4096u
__device__ __forceinline__ void cdc_add(u32 *tag, u32 *cnt, u32 *__restrict__ gcnt, u32 key, u32 v)
{
    const u32 slot = key & (CQ_CACHE - 1u);
    u32 t = tag[slot];
    if (t == 0xFFFFFFFFu) { const u32 old = atomicCAS(&tag[slot], 0xFFFFFFFFu, key); t = (old == 0xFFFFFFFFu) ? key : old; }
    if (t == key) atomicAdd(&cnt[slot], v);
    else atomicAdd(&gcnt[key], v);
}
__global__ __launch_bounds__(256) void k_cdc_count(const u8 *__restrict__ in, const u8 *__restrict__ map, CdcModel m,
                                                   u32 step, u32 *__restrict__ cnt)
{
    __shared__ u8 smap[256];
    __shared__ u32 ctag[CQ_CACHE], ccnt[CQ_CACHE];
    smap[threadIdx.x] = map[threadIdx.x];
    for (u32 j = threadIdx.x; j < CQ_CACHE; j += 256) { ctag[j] = 0xFFFFFFFFu; ccnt[j] = 0; }
    __syncthreads();
    for (u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * step; g < m.nseg; g += (u64)gridDim.x * blockDim.x * step) {
        const u64 b = g * m.seg, e = (b + m.seg < m.n) ? b + m.seg : m.n;
        u32 ctx = 0, lastKey = 0xFFFFFFFFu, run = 0;
        u64 win = 0;
#define CDC_COUNT_ONE(byte)                                                                                  \
        {                                                                                                      \
            const u32 s = smap[(u8)(byte)];                                                                    \
            const u32 key = ctx * m.A + s;                                                                     \
            if (key == lastKey) run++;                                                                         \
            else { if (run) cdc_add(ctag, ccnt, cnt, lastKey, run); lastKey = key; run = 1; }                  \
            ctx = cdc_next_ctx(ctx, s, win, m);                                                                \
        }
        u64 i = b;
        if ((((u64)in + b) & 15u) == 0)
            for (; i + 16 <= e; i += 16) {                         // 16 symbols per memory request
                const ulonglong2 v = *(const ulonglong2 *)(in + i);
#pragma unroll
                for (int j = 0; j < 8; j++) CDC_COUNT_ONE(v.x >> (8 * j));
#pragma unroll
                for (int j = 0; j < 8; j++) CDC_COUNT_ONE(v.y >> (8 * j));
            }
        for (; i < e; i++) CDC_COUNT_ONE(in[i]);
#undef CDC_COUNT_ONE
        if (run) cdc_add(ctag, ccnt, cnt, lastKey, run);
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < CQ_CACHE; j += 256)
        if (ccnt[j]) atomicAdd(&cnt[ctag[j]], ccnt[j]);
}

// fc[context * A + symbol] = frequency | cumulative frequency << 16: one look-up per symbol.  Emitted bytes are collected
// eight at a time (the stream grows towards lower addresses: the oldest byte of a group is its most significant one).
__global__ __launch_bounds__(256) void k_cdc_encode(const u8 *__restrict__ in, const u8 *__restrict__ map, CdcModel m,
                                                    const u32 *__restrict__ fc, u8 *__restrict__ scratch, u32 *__restrict__ segBytes)
{
    __shared__ u8 smap[256];
    smap[threadIdx.x] = map[threadIdx.x];
    __syncthreads();
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < m.nseg; g += (u64)gridDim.x * blockDim.x) {
        const u64 b = g * m.seg, e = (b + m.seg < m.n) ? b + m.seg : m.n;
        // context in front of the last symbol
        u32 ctx = 0;
        if (m.k) {
            const u64 len = e - b;
            for (u64 i = (len - 1 > m.k ? e - 1 - m.k : b); i + 1 < e; i++) ctx = (ctx * m.A + smap[in[i]]) % m.top;
        }
        u8 *const slotEnd = scratch + (g + 1) * (u64)CQ_SLOT(m.seg);
        u8 *q = slotEnd;
        u64 acc = 0;
        u32 nacc = 0;
        u32 x = CQ_L;
        u32 s = smap[in[e - 1]];
#define CDC_CODE_ONE()                                                                                         \
        {                                                                                                      \
            const u32 t = fc[(u64)ctx * m.A + s];                                                              \
            const u32 f = t & 0xFFFFu, c0 = t >> 16;                                                           \
            const u32 xmax = ((CQ_L >> CQ_SCALE) << 8) * f;                                                    \
            while (x >= xmax) {                                                                                \
                acc = (acc << 8) | (x & 0xFFu); x >>= 8;                                                       \
                if (++nacc == 8) { q -= 8; *(u64 *)q = acc; nacc = 0; }                                        \
            }                                                                                                  \
            const u32 d = x / f;                                                                               \
            x = (d << CQ_SCALE) + (x - d * f) + c0;                                                            \
        }
        if (((e - b) & 15u) == 0 && (((u64)in + b) & 15u) == 0) {
            // whole 16-symbol groups: the input arrives as one 16-byte load per group (one request instead of 32 byte
            // loads -- with a segment per lane every byte load is its own L2 request); w[0..1] = the group in front
            // (the symbols that leave the context live there), w[2..3] = this group
            u64 w[4];
            {
                const ulonglong2 v = *(const ulonglong2 *)(in + e - 16);
                w[2] = v.x; w[3] = v.y;
            }
            for (u64 cb = e - 16;; cb -= 16) {
                if (cb > b) { const ulonglong2 v = *(const ulonglong2 *)(in + cb - 16); w[0] = v.x; w[1] = v.y; }
                else { w[0] = 0; w[1] = 0; }
#pragma unroll
                for (int j = 15; j >= 0; j--) {
                    CDC_CODE_ONE();
                    const u64 i = cb + (u64)j;
                    if (i > b) {
                        const u32 tp = 16u + (u32)j - 1u;                                  // position of symbol i-1 in the window
                        const u32 prev = smap[(u8)(w[tp >> 3] >> (8u * (tp & 7u)))];
                        if (m.k) {
                            const u32 ti = tp - m.k;                                       // ... and of symbol i-1-k (wave-uniform)
                            const u64 wi = (ti >> 3) == 0 ? w[0] : (ti >> 3) == 1 ? w[1] : (ti >> 3) == 2 ? w[2] : w[3];
                            const u32 incoming = (i - 1 >= b + m.k) ? smap[(u8)(wi >> (8u * (ti & 7u)))] : 0u;
                            ctx = (ctx + incoming * m.top - prev) / m.A;
                        }
                        s = prev;
                    }
                }
                if (cb == b) break;
                w[2] = w[0]; w[3] = w[1];
            }
        } else {
            for (u64 i = e; i-- > b;) {
                CDC_CODE_ONE();
                if (i > b) {                                       // the symbol in front, and the context in front of it
                    const u32 prev = smap[in[i - 1]];
                    if (m.k) {
                        const u32 incoming = (i - 1 >= b + m.k) ? smap[in[i - 1 - m.k]] : 0u;
                        ctx = (ctx + incoming * m.top - prev) / m.A;
                    }
                    s = prev;
                }
            }
        }
#undef CDC_CODE_ONE
        while (nacc) { nacc--; *--q = (u8)(acc >> (8u * nacc)); }   // the oldest of the pending bytes first
        q -= 4;
        q[0] = (u8)x; q[1] = (u8)(x >> 8); q[2] = (u8)(x >> 16); q[3] = (u8)(x >> 24);
        segBytes[g] = (u32)(slotEnd - q);
    }
}

// one wavefront per segment: its stream from the end of its scratch slot to its place in the output
__global__ __launch_bounds__(256) void k_cdc_pack(const u8 *__restrict__ scratch, const u32 *__restrict__ segBytes,
                                                  const u64 *__restrict__ off, u32 nseg, u32 seg, u8 *__restrict__ out)
{
    const u32 lane = bfq_lane();
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; g < nseg; g += nwaves) {
        const u32 bytes = segBytes[g];
        const u8 *src = scratch + (g + 1) * (u64)CQ_SLOT(seg) - bytes;
        u8 *dst = out + off[g];
        for (u32 j = lane; j < bytes; j += 64) dst[j] = src[j];
    }
}

// The generic decoder: any alphabet; the symbol of a slot by binary search in the context's cumulative row.
__global__ __launch_bounds__(256) void k_cdc_decode(const u8 *__restrict__ pay, const u64 *__restrict__ off,
                                                    const u32 *__restrict__ segBytes, const u8 *__restrict__ alphabet, CdcModel m,
                                                    const u16 *__restrict__ freq, const u16 *__restrict__ cum,
                                                    u8 *__restrict__ out, u32 *__restrict__ bad)
{
    __shared__ u8 salpha[256];
    salpha[threadIdx.x] = alphabet[threadIdx.x];
    __syncthreads();
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < m.nseg; g += (u64)gridDim.x * blockDim.x) {
        const u8 *q = pay + off[g], *const qe = q + segBytes[g];
        u32 x = (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16) | ((u32)q[3] << 24);
        q += 4;
        const u64 b = g * m.seg, e = (b + m.seg < m.n) ? b + m.seg : m.n;
        u32 ctx = 0;
        u64 win = 0;
        bool ok = true;
        for (u64 i = b; i < e && ok; i++) {
            const u32 slot = x & ((1u << CQ_SCALE) - 1u);
            const u16 *fr = freq + (u64)ctx * m.A, *cu = cum + (u64)ctx * m.A;
            u32 lo = 0, hi = m.A;                                  // every row gives every symbol a share: cum is strictly increasing
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if ((u32)cu[mid] <= slot) lo = mid; else hi = mid; }
            const u32 s = lo;
            x = fr[s] * (x >> CQ_SCALE) + slot - cu[s];
            while (x < CQ_L) { if (q >= qe) { ok = false; break; } x = (x << 8) | *q++; }
            out[i] = salpha[s];
            ctx = cdc_next_ctx(ctx, s, win, m);
        }
        if (!ok) atomicAdd(bad, 1u);
    }
}

// Alphabets of up to 8 symbols (the DNA stream, binned qualities): a context's whole row is ONE 16-byte load (rows padded to
// 8 entries), the stream is read 8 bytes at a time and the output leaves 8 symbols at a time -- with a segment per lane
// every narrow access is its own L2 request, and those requests, not the arithmetic, were the decoder's time.
__global__ __launch_bounds__(256) void k_cdc_decode8(const u8 *__restrict__ pay, const u64 *__restrict__ off,
                                                     const u32 *__restrict__ segBytes, const u8 *__restrict__ alphabet, CdcModel m,
                                                     const u16 *__restrict__ freq8, u8 *__restrict__ out, u32 *__restrict__ bad)
{
    __shared__ u8 salpha[256];
    salpha[threadIdx.x] = alphabet[threadIdx.x];
    __syncthreads();
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < m.nseg; g += (u64)gridDim.x * blockDim.x) {
        const u8 *q = pay + off[g];
        const u32 nbytes = segBytes[g];
        u32 x = (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16) | ((u32)q[3] << 24);
        if (x < CQ_L) { atomicAdd(bad, 1u); continue; }             // an encoder's final state is never below the renormalisation bound
        u32 used = 4;                                              // bytes of the stream consumed so far
        u64 ib = 0;                                                // bytes not yet consumed, the next one in the low byte
        u32 ni = 0;
        const u64 b = g * m.seg, e = (b + m.seg < m.n) ? b + m.seg : m.n;
        u32 ctx = 0;
        u64 win = 0, ob = 0;
        bool ok = true;
        for (u64 i = b; i < e; i++) {
            const u32 slot = x & ((1u << CQ_SCALE) - 1u);
            const uint4 row = *(const uint4 *)(freq8 + (u64)ctx * 8);
            const u32 fw[4] = {row.x, row.y, row.z, row.w};
            u32 s = 0, c0 = 0, f = fw[0] & 0xFFFFu;
#pragma unroll
            for (int t = 1; t < 8; t++) {                          // first symbol whose share reaches past the slot
                const u32 ft = (t & 1) ? fw[t >> 1] >> 16 : fw[t >> 1] & 0xFFFFu;
                const bool past = slot >= c0 + f;
                c0 = past ? c0 + f : c0; s = past ? (u32)t : s; f = past ? ft : f;
            }
            x = f * (x >> CQ_SCALE) + slot - c0;
            while (x < CQ_L) {
                // a damaged stream reads zeros past its segment, not memory (the buffer is padded by 16 bytes) -- and only a
                // few of them: a state that stays 0 would otherwise refill for ever (a zero-filled page inside a payload)
                if (used >= nbytes + 4) { ok = false; break; }
                if (ni == 0) { ib = (used < nbytes) ? *(const u64 *)(q + used) : 0ull; ni = 8; }
                x = (x << 8) | (u32)(ib & 0xFFu); ib >>= 8; ni--; used++;
            }
            if (!ok) break;
            ob |= (u64)salpha[s] << (8u * (u32)((i - b) & 7u));
            if (((i - b) & 7u) == 7u) { *(u64 *)(out + i - 7) = ob; ob = 0; }
            ctx = cdc_next_ctx(ctx, s, win, m);
        }
        for (u64 i = e - ((e - b) & 7u); i < e; i++) out[i] = (u8)(ob >> (8u * (u32)((i - b) & 7u)));   // a last segment that is no multiple of 8
        if (used > nbytes) ok = false;
        if (!ok) atomicAdd(bad, 1u);
    }
}

// checksum of the raw stream (oracle/bfq_codec_ref.c states it): a sum of position-keyed terms, one 64-bit word per lane
#define CQ_HDR 44u                                              // magic, raw_len, five u32, checksum
__host__ __device__ static inline u64 cdc_mix64(u64 z) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z; }
__global__ __launch_bounds__(256) void k_cdc_checksum(const u8 *__restrict__ in, u64 n, u64 *__restrict__ out)
{
    const u64 nw = (n + 7) / 8;
    u64 sum = 0;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < nw; j += (u64)gridDim.x * blockDim.x) {
        u64 w = 0;
        if (8 * j + 8 <= n) __builtin_memcpy(&w, in + 8 * j, 8);   // any alignment
        else for (u64 b = 0; 8 * j + b < n; b++) w |= (u64)in[8 * j + b] << (8 * b);
        sum += cdc_mix64(w + (j + 1) * 0x9E3779B97F4A7C15ull);
    }
    sum = bfq_readlane64(bfq_wave_incscan64(sum), 63);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd((unsigned long long *)out, (unsigned long long)sum);
}
// d_tmp: 8 bytes of device scratch
u64 bfq_codec_checksum_device(bfq_ctx *c, const u8 *d_in, u64 n, u64 *d_tmp)
{
    HIP_CHECK(hipMemsetAsync(d_tmp, 0, 8, c->stream));
    if (n) KLAUNCH(c, K_CODEC, (double)n, k_cdc_checksum, bfq_grid((n + 7) / 8, 256 * 16), 256, d_in, n, d_tmp);
    u64 sum = 0;
    HIP_CHECK(hipMemcpyAsync(&sum, d_tmp, 8, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    return cdc_mix64(n ^ sum);
}

// ---- host side ----------------------------------------------------------------------------------------------
static u32 cdc_choose_k(u32 A, u64 n)
{
    u64 limit = n >> 4;
    if (limit < 4096) limit = 4096;
    if (limit > CQ_MAX_TABLE) limit = CQ_MAX_TABLE;
    u32 k = 0;
    u64 p = (u64)A * A;
    while (k < 8 && p <= limit) { k++; p *= A; }
    return k;
}
static void cdc_normalise(const u32 *cnt, u32 A, u16 *f)
{
    const u32 M = 1u << CQ_SCALE;
    u64 T = 0;
    for (u32 s = 0; s < A; s++) T += cnt[s];
    u32 sum = 0;
    for (u32 s = 0; s < A; s++) {
        u32 v = T ? (u32)(((u64)cnt[s] * M) / T) : 0;
        if (v == 0) v = 1;                                        // every symbol of the alphabet can be coded in every context
        f[s] = (u16)v; sum += v;
    }
    while (sum > M) {
        u32 best = 0;
        for (u32 s = 1; s < A; s++) if (f[s] > f[best]) best = s;
        u32 d = sum - M;
        if (d > (u32)f[best] - 1u) d = (u32)f[best] - 1u;
        f[best] = (u16)(f[best] - d); sum -= d;
    }
    if (sum < M) {
        u32 best = 0;
        for (u32 s = 1; s < A; s++) if (f[s] > f[best]) best = s;
        f[best] = (u16)(f[best] + (M - sum));
    }
}
// (12 - log2 f) * 256 for a frequency f in 1 .. 4096, in integers (oracle/bfq_codec_ref.c: bit_cost)
static u32 cdc_bit_cost(u32 f)
{
    u32 e = 0;
    while ((2u << e) <= f) e++;
    u64 m = (u64)f << (31 - e);
    u32 frac = 0;
    for (int i = 0; i < 8; i++) {
        m = (m * m) >> 31;
        frac <<= 1;
        if (m >> 32) { frac |= 1; m >>= 1; }
    }
    return CQ_SCALE * 256 - (e * 256 + frac);
}
// The order the container is made with (oracle/bfq_codec_ref.c: choose_order): the counts were taken at order kmax; for every
// order <= kmax the container's size is estimated (payload from the rows' normalised frequencies, times the sample step, + the
// table) and the smallest wins.  cnt is left holding the counts of the chosen order.
static u32 cdc_choose_order(std::vector<u32> &cnt, u32 A, u32 kmax, u32 S)
{
    u32 best = kmax;
    u64 bestBits = ~0ull, nctx = 1;
    for (u32 j = 0; j < kmax; j++) nctx *= A;
    std::vector<u32> lvl(cnt.begin(), cnt.begin() + nctx * A), keep;
    u16 f[256];
    for (u32 k = kmax;; k--) {
        u64 bits = 0, used = 0;
        for (u64 x = 0; x < nctx; x++) {
            u64 T = 0;
            for (u32 s = 0; s < A; s++) T += lvl[x * A + s];
            if (!T) continue;
            used++;
            cdc_normalise(lvl.data() + x * A, A, f);
            for (u32 s = 0; s < A; s++) bits += (u64)lvl[x * A + s] * cdc_bit_cost(f[s]);
        }
        bits = bits / 256 * S + used * A * 16 + nctx;
        if (bits < bestBits) { bestBits = bits; best = k; keep.assign(lvl.begin(), lvl.begin() + nctx * A); }
        if (k == 0) break;
        const u64 low = nctx / A;
        for (u64 x = low; x < nctx; x++)
            for (u32 s = 0; s < A; s++) {
                const u64 v = (u64)lvl[(x % low) * A + s] + lvl[x * A + s];
                lvl[(x % low) * A + s] = v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)v;
            }
        nctx = low;
    }
    std::copy(keep.begin(), keep.end(), cnt.begin());
    return best;
}
static u32 cdc_choose_seg(u64 n)
{
    u32 seg = CQ_SEG_MAX;
    while (seg > 1024 && n / seg < 65536) seg >>= 1;
    return seg;
}
static u32 cdc_sample_step(u64 n)
{
    const u64 S = n >> 24;                                        // at least 16 M symbols are counted (all of a shorter stream)
    return S < 1 ? 1u : (S > 64 ? 64u : (u32)S);
}
static void put32(u8 *p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); p[3] = (u8)(v >> 24); }
static void put64(u8 *p, u64 v) { put32(p, (u32)v); put32(p + 4, (u32)(v >> 32)); }
static u32 get32(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
static u64 get64(const u8 *p) { return (u64)get32(p) | ((u64)get32(p + 4) << 32); }

// upper bound of a container for n raw bytes (what callers size their output buffers with)
u64 bfq_codec_bound(u64 n)
{
    const u64 nseg = (n + 1023) / 1024;
    // (+ n / 2: a BFQDNAC1 container holds a second, small BFQRANS2 container -- the line lengths -- in front of its payload)
    return 32 + CQ_HDR + 256 + 512 + CQ_MAX_TABLE / 8 + 2ull * CQ_MAX_TABLE + 4 * nseg + n + n / 2 + 8 * nseg + 64 + n / 2 + (16u << 20);
}
u64 bfq_dnac_workspace(u64 n);                                                                   // k_dnac.hip
u64 bfq_dnac_compress_device(bfq_ctx *c, const u8 *d_in, u64 n, u8 *d_out, u64 cap, u64 *nbases);
u64 bfq_dnac_decompress_device(bfq_ctx *c, const u8 *h_in, const u8 *d_in, u64 len, u8 *d_out, u64 cap);
u64 bfq_dnac_member_len(const u8 *h_in, u64 len);
// device workspace of one compress / decompress call
u64 bfq_codec_workspace(u64 n)
{
    const u64 nseg = (n + 1023) / 1024;
    // + the line-delta transform of streams with 8 .. 128 bytes per line: line index, prefix lengths, record sizes / offsets, the records
    // + the context table of the read-order DNA container (k_dnac.hip; its per-base arrays fit what the line transform counts)
    return n + bfq_codec_bound(n) + nseg * (u64)CQ_SLOT(1024) + 16ull * CQ_MAX_TABLE + 24 * nseg + 4 * n + (64u << 20) + bfq_dnac_workspace(n);
}

// d_in: n raw bytes on the device.  The container goes to d_out (capacity cap); returns its length.
// dry: only the container's length is wanted (nothing is written to d_out)
u64 bfq_rans_compress_device(bfq_ctx *c, const u8 *d_in, u64 n, u8 *d_out, u64 cap, bool dry)
{
    const size_t mk = c->mark();
    const u64 checksum = bfq_codec_checksum_device(c, d_in, n, c->alloc<u64>(1));
    u32 *d_present = c->alloc<u32>(256);
    HIP_CHECK(hipMemsetAsync(d_present, 0, 1024, c->stream));
    if (n) KLAUNCH(c, K_CODEC, (double)n, k_cdc_present, bfq_grid(n, 256 * 64), 256, d_in, n, d_present);
    u32 present[256];
    HIP_CHECK(hipMemcpyAsync(present, d_present, 1024, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    u8 alphabet[256] = {0}, map[256] = {0};
    u32 A = 0;
    for (u32 b = 0; b < 256; b++) if (present[b]) { map[b] = (u8)A; alphabet[A++] = (u8)b; }
    if (A == 0) A = 1;
    CdcModel m;
    const u32 step = cdc_sample_step(n);
    m.A = A; m.k = cdc_choose_k(A, n / step); m.n = n; m.seg = cdc_choose_seg(n); m.nseg = (u32)((n + m.seg - 1) / m.seg);
    u64 nctx = 1;
    for (u32 j = 0; j < m.k; j++) nctx *= A;
    m.top = (u32)nctx;
    u64 E = nctx * A;
    u8 *d_map = c->alloc<u8>(256);
    HIP_CHECK(hipMemcpyAsync(d_map, map, 256, hipMemcpyHostToDevice, c->stream));
    u32 *d_cnt = c->alloc<u32>(E);
    HIP_CHECK(hipMemsetAsync(d_cnt, 0, 4 * E, c->stream));
    if (n) KLAUNCH(c, K_CODEC, (double)n / step, k_cdc_count, bfq_grid((m.nseg + step - 1) / step, 256), 256, d_in, (const u8 *)d_map, m, step, d_cnt);
    std::vector<u32> cnt(E);
    HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, 4 * E, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    m.k = cdc_choose_order(cnt, A, m.k, step);                     // the counts were taken at the highest order considered
    nctx = 1;
    for (u32 j = 0; j < m.k; j++) nctx *= A;
    m.top = (u32)nctx;
    E = nctx * A;
    std::vector<u16> freq(E, 0), cum(E, 0);
    std::vector<u8> used((nctx + 7) / 8, 0);
    u64 nused = 0;
    u32 cnt0[256] = {0};
    u16 dflt[256] = {0};
    for (u64 x = 0; x < nctx; x++)
        for (u32 s = 0; s < A; s++) { const u64 v = (u64)cnt0[s] + cnt[x * A + s]; cnt0[s] = v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)v; }
    cdc_normalise(cnt0, A, dflt);
    for (u64 x = 0; x < nctx; x++) {
        u64 T = 0;
        for (u32 s = 0; s < A; s++) T += cnt[x * A + s];
        if (T) {
            nused++;
            used[x >> 3] |= (u8)(1u << (x & 7));
            cdc_normalise(cnt.data() + x * A, A, freq.data() + x * A);
        } else memcpy(freq.data() + x * A, dflt, 2 * A);
        u32 acc = 0;
        for (u32 s = 0; s < A; s++) { cum[x * A + s] = (u16)acc; acc += freq[x * A + s]; }
    }
    const u64 hdr = CQ_HDR + 256 + 2ull * A + used.size() + nused * A * 2 + 4ull * m.nseg;
    if (hdr > cap && !dry) throw BfqError{BFQ_E_ARG, "output buffer too small for the compressed stream"};
    std::vector<u32> fcv(E);
    for (u64 x = 0; x < E; x++) fcv[x] = (u32)freq[x] | ((u32)cum[x] << 16);
    u32 *d_fc = c->alloc<u32>(E);
    HIP_CHECK(hipMemcpyAsync(d_fc, fcv.data(), 4 * E, hipMemcpyHostToDevice, c->stream));
    u32 *d_segBytes = c->alloc<u32>(m.nseg + 1);
    u64 *d_off = c->alloc<u64>(m.nseg + 1), *d_total = c->alloc<u64>(1);
    u8 *scratch = c->alloc<u8>((u64)m.nseg * CQ_SLOT(m.seg) + 16);
    u64 total = 0;
    std::vector<u32> segBytes(m.nseg);
    if (m.nseg) {
        KLAUNCH(c, K_CODEC, 3.0 * (double)n, k_cdc_encode, bfq_grid(m.nseg, 256), 256, d_in, (const u8 *)d_map, m, (const u32 *)d_fc,
                scratch, d_segBytes);
        bfq_exscan_u32(c, d_segBytes, d_off, m.nseg, d_total);
        HIP_CHECK(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipMemcpyAsync(segBytes.data(), d_segBytes, 4ull * m.nseg, hipMemcpyDeviceToHost, c->stream));
        c->sync();
        if (dry) { c->release(mk); return hdr + total; }
        if (hdr + total > cap) throw BfqError{BFQ_E_ARG, "output buffer too small for the compressed stream"};
        KLAUNCH(c, K_CODEC, 2.0 * (double)total, k_cdc_pack, bfq_grid((u64)m.nseg * 64, 256), 256, (const u8 *)scratch, (const u32 *)d_segBytes,
                (const u64 *)d_off, m.nseg, m.seg, d_out + hdr);
    }
    if (dry) { c->release(mk); return hdr + total; }
    std::vector<u8> h(hdr);
    u8 *p = h.data();
    memcpy(p, "BFQRANS2", 8); put64(p + 8, n);
    put32(p + 16, m.seg); put32(p + 20, m.nseg); put32(p + 24, A); put32(p + 28, m.k); put32(p + 32, CQ_SCALE);
    put64(p + 36, checksum);
    memcpy(p + CQ_HDR, alphabet, 256);
    for (u32 s = 0; s < A; s++) { p[CQ_HDR + 256 + 2 * s] = (u8)dflt[s]; p[CQ_HDR + 256 + 2 * s + 1] = (u8)(dflt[s] >> 8); }
    memcpy(p + CQ_HDR + 256 + 2ull * A, used.data(), used.size());
    u8 *rows = p + CQ_HDR + 256 + 2ull * A + used.size();
    for (u64 x = 0; x < nctx; x++) {
        if (!((used[x >> 3] >> (x & 7)) & 1)) continue;
        for (u32 s = 0; s < A; s++) { rows[0] = (u8)freq[x * A + s]; rows[1] = (u8)(freq[x * A + s] >> 8); rows += 2; }
    }
    for (u32 g = 0; g < m.nseg; g++) put32(rows + 4ull * g, segBytes[g]);
    HIP_CHECK(hipMemcpyAsync(d_out, h.data(), hdr, hipMemcpyHostToDevice, c->stream));
    c->sync();
    c->release(mk);
    return hdr + total;
}

// h_hdr: the first bytes of a container (at least min(len, bfq_codec_header_bound()) of them) on the host.
// Parses the header; returns its length and the raw length.
struct CdcHeader { CdcModel m; u64 hdr; std::vector<u8> alphabet; std::vector<u16> freq, cum; std::vector<u32> segBytes; };
static void cdc_parse(const u8 *in, u64 len, CdcHeader &H)
{
    const BfqError bad{BFQ_E_ARG, "not a BFQRANS2 stream (or a damaged one)"};
    if (len < CQ_HDR + 256 || memcmp(in, "BFQRANS2", 8)) throw bad;
    CdcModel &m = H.m;
    m.n = get64(in + 8);
    const u32 seg = get32(in + 16), scale = get32(in + 32);
    m.seg = seg;
    m.nseg = get32(in + 20); m.A = get32(in + 24); m.k = get32(in + 28);
    if (seg != cdc_choose_seg(m.n) || scale != CQ_SCALE || m.A == 0 || m.A > 256 || m.k > 8 || m.nseg != (m.n + seg - 1) / seg) throw bad;
    u64 nctx = 1;
    for (u32 j = 0; j < m.k; j++) { nctx *= m.A; if (nctx > CQ_MAX_TABLE) throw bad; }
    if (nctx * m.A > CQ_MAX_TABLE) throw bad;
    m.top = (u32)nctx;
    H.alphabet.assign(in + CQ_HDR, in + CQ_HDR + 256);
    if (CQ_HDR + 256 + 2ull * m.A + (nctx + 7) / 8 > len) throw bad;
    const u8 *dfl = in + CQ_HDR + 256;
    const u8 *used = dfl + 2ull * m.A;
    const u8 *rows = used + (nctx + 7) / 8;
    H.freq.assign(nctx * m.A, 0); H.cum.assign(nctx * m.A, 0);
    for (u64 x = 0; x < nctx; x++) {
        const u8 *row = dfl;
        if ((used[x >> 3] >> (x & 7)) & 1) {
            if ((u64)(rows - in) + 2ull * m.A > len) throw bad;
            row = rows; rows += 2ull * m.A;
        }
        u32 acc = 0;
        for (u32 s = 0; s < m.A; s++) { H.freq[x * m.A + s] = (u16)(row[2 * s] | (row[2 * s + 1] << 8)); H.cum[x * m.A + s] = (u16)acc; acc += H.freq[x * m.A + s]; }
        if (acc != (1u << CQ_SCALE)) throw bad;
    }
    if ((u64)(rows - in) + 4ull * m.nseg > len) throw bad;
    H.segBytes.resize(m.nseg);
    u64 total = 0;
    for (u32 g = 0; g < m.nseg; g++) { H.segBytes[g] = get32(rows + 4ull * g); if (H.segBytes[g] < 4) throw bad; total += H.segBytes[g]; }
    H.hdr = (u64)(rows - in) + 4ull * m.nseg;
    if (H.hdr + total > len) throw bad;
}

// A file may hold several containers back to back (one per block of a sharded run: bfqzip_amd/parallel.py).
// Length of the first member / raw length of all members.
u64 bfq_codec_member_len(const u8 *h_in, u64 len)
{
    if (len >= 8 && !memcmp(h_in, "BFQDNAC1", 8)) return bfq_dnac_member_len(h_in, len);
    if (len >= 32 && !memcmp(h_in, "BFQLINE1", 8)) return 32 + bfq_codec_member_len(h_in + 32, len - 32);
    CdcHeader H;
    cdc_parse(h_in, len, H);
    u64 total = H.hdr;
    for (u32 g = 0; g < H.m.nseg; g++) total += H.segBytes[g];
    return total;
}
u64 bfq_codec_raw_len(const u8 *h_in, u64 len)
{
    u64 pos = 0, raw = 0;
    do {
        const bool lx = (len - pos >= 32 && !memcmp(h_in + pos, "BFQLINE1", 8)) || (len - pos >= 72 && !memcmp(h_in + pos, "BFQDNAC1", 8));
        if (!lx && (len - pos < CQ_HDR + 256 || memcmp(h_in + pos, "BFQRANS2", 8))) throw BfqError{BFQ_E_ARG, "not a BFQRANS2 stream"};
        raw += get64(h_in + pos + 8);
        pos += bfq_codec_member_len(h_in + pos, len - pos);
    } while (pos < len);
    return raw;
}

// h_in: the whole container on the host (its header is parsed there), d_in: the same bytes on the device.
// The raw bytes go to d_out (capacity cap); returns their number.
u64 bfq_rans_decompress_device(bfq_ctx *c, const u8 *h_in, const u8 *d_in, u64 len, u8 *d_out, u64 cap)
{
    CdcHeader H;
    cdc_parse(h_in, len, H);
    const CdcModel &m = H.m;
    if (m.n > cap) throw BfqError{BFQ_E_ARG, "output buffer too small for the raw stream"};
    if (!m.nseg) return 0;
    const size_t mk = c->mark();
    const u64 E = (u64)m.top * m.A;
    u16 *d_freq = c->alloc<u16>(E), *d_cum = c->alloc<u16>(E);
    u8 *d_alpha = c->alloc<u8>(256);
    u32 *d_segBytes = c->alloc<u32>(m.nseg + 1), *d_bad = c->alloc<u32>(1);
    u64 *d_off = c->alloc<u64>(m.nseg + 1);
    HIP_CHECK(hipMemcpyAsync(d_freq, H.freq.data(), 2 * E, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(d_cum, H.cum.data(), 2 * E, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(d_alpha, H.alphabet.data(), 256, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(d_segBytes, H.segBytes.data(), 4ull * m.nseg, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, c->stream));
    bfq_exscan_u32(c, d_segBytes, d_off, m.nseg, nullptr);
    if (m.A <= 8) {                                                // rows padded to 8 entries: one aligned 16-byte load per symbol
        std::vector<u16> f8((u64)m.top * 8, 0);
        for (u64 x = 0; x < m.top; x++)
            for (u32 t = 0; t < m.A; t++) f8[x * 8 + t] = H.freq[x * m.A + t];
        u16 *d_f8 = c->alloc<u16>((u64)m.top * 8);
        HIP_CHECK(hipMemcpyAsync(d_f8, f8.data(), 16ull * m.top, hipMemcpyHostToDevice, c->stream));
        KLAUNCH(c, K_CODEC, 3.0 * (double)m.n, k_cdc_decode8, bfq_grid(m.nseg, 256), 256, d_in + H.hdr, (const u64 *)d_off, (const u32 *)d_segBytes,
                (const u8 *)d_alpha, m, (const u16 *)d_f8, d_out, d_bad);
        c->sync();                                                 // f8 is a host temporary
    } else
        KLAUNCH(c, K_CODEC, 3.0 * (double)m.n, k_cdc_decode, bfq_grid(m.nseg, 256), 256, d_in + H.hdr, (const u64 *)d_off, (const u32 *)d_segBytes,
                (const u8 *)d_alpha, m, (const u16 *)d_freq, (const u16 *)d_cum, d_out, d_bad);
    u32 bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    const u64 sum = bad ? 0 : bfq_codec_checksum_device(c, d_out, m.n, c->alloc<u64>(1));
    c->release(mk);
    if (bad) throw BfqError{BFQ_E_ARG, "damaged BFQRANS2 stream"};
    if (sum != get64(h_in + 36)) throw BfqError{BFQ_E_ARG, "damaged BFQRANS2 stream (checksum of the decoded bytes)"};
    return m.n;
}

// ---- line-delta transform of line-structured streams (read names): oracle/bfq_codec_ref.c states it ---------------------
#define CQ_LINE_R 256u
// p[i] = bytes line i shares with the line before (0 for every R-th line; at most 254 and the line's length - 1),
// sizes[i] = 1 + length - p[i] = bytes of its record.  One lane per line, 8 bytes per comparison.
__global__ __launch_bounds__(256) void k_lx_sizes(const u8 *__restrict__ in, const u64 *__restrict__ lineEnd, u64 nl,
                                                  u8 *__restrict__ pfx, u32 *__restrict__ sizes)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += (u64)gridDim.x * blockDim.x) {
        const u64 s = i ? lineEnd[i - 1] + 1 : 0, len = lineEnd[i] - s + 1;
        u64 p = 0;
        if (i % CQ_LINE_R) {
            const u64 ps = i >= 2 ? lineEnd[i - 2] + 1 : 0, plen = s - ps;
            u64 lim = len - 1;
            if (lim > plen) lim = plen;
            if (lim > 254) lim = 254;
            while (p + 8 <= lim) {
                u64 a, b;
                __builtin_memcpy(&a, in + s + p, 8); __builtin_memcpy(&b, in + ps + p, 8);
                const u64 x = a ^ b;
                if (x) { p += (u64)(__builtin_ctzll(x) >> 3); lim = p; break; }
                p += 8;
            }
            while (p < lim && in[s + p] == in[ps + p]) p++;
        }
        pfx[i] = (u8)p;
        sizes[i] = (u32)(1 + len - p);
    }
}
__global__ __launch_bounds__(256) void k_lx_write(const u8 *__restrict__ in, const u64 *__restrict__ lineEnd, u64 nl,
                                                  const u8 *__restrict__ pfx, const u64 *__restrict__ off, u8 *__restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += (u64)gridDim.x * blockDim.x) {
        const u64 s = i ? lineEnd[i - 1] + 1 : 0, e = lineEnd[i];
        const u32 p = pfx[i];
        u8 *o = out + off[i];
        *o++ = (u8)(p + (p >= 10u ? 1u : 0u));
        for (u64 j = s + p; j <= e; j++) *o++ = in[j];
    }
}
// inverse: length of every line from its record; then the lines of a group of R one after the other (a lane per group)
__global__ __launch_bounds__(256) void k_lxi_len(const u8 *__restrict__ t, const u64 *__restrict__ recEnd, u64 nl, u32 *__restrict__ lens,
                                                 u32 *__restrict__ bad)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += (u64)gridDim.x * blockDim.x) {
        const u64 rs = i ? recEnd[i - 1] + 1 : 0;
        u32 p = t[rs];
        if (p == 10u || rs == recEnd[i]) atomicAdd(bad, 1u);      // a record starts with its prefix byte, never with the line end
        if (p > 10u) p--;
        if (p && i % CQ_LINE_R == 0) atomicAdd(bad, 1u);
        lens[i] = p + (u32)(recEnd[i] - rs);
    }
}
__global__ __launch_bounds__(256) void k_lxi_write(const u8 *__restrict__ t, const u64 *__restrict__ recEnd, u64 nl,
                                                   const u64 *__restrict__ outOff, u8 *__restrict__ out, u32 *__restrict__ bad)
{
    const u64 ngroups = (nl + CQ_LINE_R - 1) / CQ_LINE_R;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (u64)gridDim.x * blockDim.x) {
        const u64 i0 = g * CQ_LINE_R, i1 = (i0 + CQ_LINE_R < nl) ? i0 + CQ_LINE_R : nl;
        u64 prev = 0, prevLen = 0;
        for (u64 i = i0; i < i1; i++) {
            const u64 rs = i ? recEnd[i - 1] + 1 : 0, re = recEnd[i];
            u32 p = t[rs];
            if (p > 10u) p--;
            if (p > prevLen) { atomicAdd(bad, 1u); p = (u32)prevLen; }
            u8 *o = out + outOff[i];
            for (u32 j = 0; j < p; j++) o[j] = out[prev + j];
            for (u64 j = rs + 1; j <= re; j++) o[p + (j - rs - 1)] = t[j];
            prev = outOff[i]; prevLen = p + (re - rs);
        }
    }
}

// transformed stream of d_in (arena) and its length, or 0 when the transform does not apply / does not pay
static u64 line_xform_device(bfq_ctx *c, const u8 *d_in, u64 n, u8 **d_T, u64 *nlines)
{
    if (n < 2) return 0;
    u8 last = 0;
    HIP_CHECK(hipMemcpyAsync(&last, d_in + n - 1, 1, hipMemcpyDeviceToHost, c->stream));
    const u64 nl = bfq_fastq_count_lines(c, d_in, n) - 1;        // (synchronises)
    if (last != 10 || nl < 2 || n / nl < 8 || n / nl > 128) return 0;
    u64 nl2 = 0;
    const u64 *lineEnd = bfq_line_index(c, d_in, n, &nl2);
    u8 *pfx = c->alloc<u8>(nl);
    u32 *sizes = c->alloc<u32>(nl);
    u64 *off = c->alloc<u64>(nl + 1), *d_total = c->alloc<u64>(1);
    KLAUNCH(c, K_CODEC, 2.0 * (double)n, k_lx_sizes, bfq_grid(nl, 256), 256, d_in, lineEnd, nl, pfx, sizes);
    bfq_exscan_u32(c, sizes, off, nl, d_total);
    u64 total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (total * 4 > n * 3) return 0;
    *d_T = c->alloc<u8>(total + 16);
    KLAUNCH(c, K_CODEC, (double)n + (double)total, k_lx_write, bfq_grid(nl, 256), 256, d_in, lineEnd, nl, (const u8 *)pfx, (const u64 *)off, *d_T);
    *nlines = nl;
    return total;
}

// d_in: n raw bytes on the device.  The container goes to d_out (capacity cap); returns its length.
u64 bfq_codec_compress_device(bfq_ctx *c, const u8 *d_in, u64 n, u8 *d_out, u64 cap)
{
    {   // read-order DNA: its own container (k_dnac.hip); 0 = the stream is something else
        u64 nbases = 0;
        const u64 got = bfq_dnac_compress_device(c, d_in, n, d_out, cap, &nbases);
        // little coverage leaves the table nothing to learn: above 1.6 bits per base the smaller of this and the plain static container
        if (got && (5 * got <= nbases || bfq_rans_compress_device(c, d_in, n, d_out, cap, true) >= got)) return got;
        if (got) return bfq_rans_compress_device(c, d_in, n, d_out, cap, false);
    }
    const size_t mk = c->mark();
    u8 *d_T = nullptr;
    u64 nl = 0;
    const u64 xl = line_xform_device(c, d_in, n, &d_T, &nl);
    u64 got;
    if (!xl) got = bfq_rans_compress_device(c, d_in, n, d_out, cap, false);
    else {
        if (cap < 32) throw BfqError{BFQ_E_ARG, "output buffer too small for the compressed stream"};
        u8 h[32];
        memcpy(h, "BFQLINE1", 8); put64(h + 8, n); put32(h + 16, CQ_LINE_R); put32(h + 20, 0); put64(h + 24, nl);
        HIP_CHECK(hipMemcpyAsync(d_out, h, 32, hipMemcpyHostToDevice, c->stream));
        c->sync();
        got = 32 + bfq_rans_compress_device(c, d_T, xl, d_out + 32, cap - 32, false);
    }
    c->release(mk);
    return got;
}

// h_in: the whole container on the host (its header is parsed there), d_in: the same bytes on the device.
// The raw bytes go to d_out (capacity cap); returns their number.
u64 bfq_codec_decompress_device(bfq_ctx *c, const u8 *h_in, const u8 *d_in, u64 len, u8 *d_out, u64 cap)
{
    if (len >= 8 && !memcmp(h_in, "BFQDNAC1", 8)) return bfq_dnac_decompress_device(c, h_in, d_in, len, d_out, cap);
    if (len < 32 || memcmp(h_in, "BFQLINE1", 8)) return bfq_rans_decompress_device(c, h_in, d_in, len, d_out, cap);
    const BfqError bad{BFQ_E_ARG, "damaged BFQLINE1 stream"};
    const u64 n = get64(h_in + 8), nl = get64(h_in + 24);
    if (n > cap || get32(h_in + 16) != CQ_LINE_R || nl < 2 || nl > n) throw bad;
    if (len - 32 < CQ_HDR + 256 || memcmp(h_in + 32, "BFQRANS2", 8)) throw bad;
    const u64 xl = get64(h_in + 32 + 8);
    if (xl > n + nl) throw bad;
    const size_t mk = c->mark();
    u8 *d_T = c->alloc<u8>(xl + 16);
    if (bfq_rans_decompress_device(c, h_in + 32, d_in + 32, len - 32, d_T, xl) != xl) throw bad;
    u64 nrec = 0;
    const u64 *recEnd = bfq_line_index(c, d_T, xl, &nrec);
    if (nrec != nl) throw bad;
    u32 *lens = c->alloc<u32>(nl), *d_bad = c->alloc<u32>(1);
    u64 *off = c->alloc<u64>(nl + 1), *d_total = c->alloc<u64>(1);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, c->stream));
    KLAUNCH(c, K_CODEC, (double)xl, k_lxi_len, bfq_grid(nl, 256), 256, (const u8 *)d_T, recEnd, nl, lens, d_bad);
    bfq_exscan_u32(c, lens, off, nl, d_total);
    u64 total = 0;
    u32 nbad = 0;
    HIP_CHECK(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipMemcpyAsync(&nbad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (nbad || total != n) throw bad;
    KLAUNCH(c, K_CODEC, (double)xl + (double)n, k_lxi_write, bfq_grid((nl + CQ_LINE_R - 1) / CQ_LINE_R, 256), 256, (const u8 *)d_T, recEnd, nl,
            (const u64 *)off, d_out, d_bad);
    HIP_CHECK(hipMemcpyAsync(&nbad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    c->release(mk);
    if (nbad) throw bad;
    return n;
}
