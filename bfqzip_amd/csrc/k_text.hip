// k_text.hip -- reads -> terminated text of symbol codes -> 3-bit packed text ->
// (21-symbol key, payload) pairs, one per suffix.  Pure streaming kernels.
//
// Replaces the input stage of the absent step-1 tool (call site BFQzip.py:178-189):
// the text is the read collection with one terminator per read, no global end
// marker (gsufsort built with TERMINATOR=0, reference Makefile:18).
#include "bfq_internal.h"
#include "bfq_device.h"

// 16 lanes per read, 8 bytes per lane and step (unaligned 8-byte loads/stores are fine on gfx950):
// symbol codes + qualities into terminated text order
__device__ __forceinline__ u64 codes8(u64 x, bool *bad)      // 8 ASCII bases -> 8 codes
{
    // (c >> 1) & 7 separates A0 C1 T2 G3 N7; nibble LUTs give the code and the letter to verify against
    u64 out = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        u32 c = (u32)(x >> (8 * k)) & 0xFFu;
        u32 h = (c >> 1) & 7u;
        u32 code = (0x40003521u >> (4 * h)) & 0xFu;                       // h: 0 1 2 3 . . . 7 -> 1 2 5 3 . . . 4
        u32 want = (u32)((0x4E00000047544341ull >> (8 * h)) & 0xFFu);     // 'A' 'C' 'T' 'G' . . . 'N'
        if (c != want || code == 0) { *bad = true; code = 4; }
        out |= (u64)code << (8 * k);
    }
    return out;
}

__global__ __launch_bounds__(256) void k_text_from_reads(const u8 *__restrict__ bases, const u8 *__restrict__ quals,
                                                         const u64 *__restrict__ roff, u64 N, u8 *__restrict__ T8,
                                                         u8 *__restrict__ Q8, DevCounters *cnt)
{
    const u32 sub = threadIdx.x & 15u;
    u64 grp = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const u64 ngrp = ((u64)gridDim.x * blockDim.x) >> 4;
    bool bad = false, tooLong = false;
    for (u64 i = grp; i < N; i += ngrp) {
        u64 b = roff[i], e = roff[i + 1];
        u64 t0 = b + i, len = e - b;
        if (len > BFQ_MAX_READ_LEN) tooLong = true;
        if (len >= 8) {                                           // the last word overlaps the one before: no byte tail
            const u64 nw = (len + 7) >> 3;
            for (u64 j = sub; j < nw; j += 16) {
                const u64 k = (8 * j + 8 <= len) ? 8 * j : len - 8;
                u64 xb = *(const u64 *)(bases + b + k);
                u64 xq = *(const u64 *)(quals + b + k);
                *(u64 *)(T8 + t0 + k) = codes8(xb, &bad);
                *(u64 *)(Q8 + t0 + k) = xq;
            }
        } else if (sub == 0) {
            for (u64 j = 0; j < len; j++) {
                u32 code = bfq_base_code(bases[b + j]);
                if (code == BFQ_CODE_INVALID) { bad = true; code = 4; }
                T8[t0 + j] = (u8)code;
                Q8[t0 + j] = quals[b + j];
            }
        }
        if (sub == 0) { T8[t0 + len] = 0; Q8[t0 + len] = (u8)'#'; }
    }
    if (bad) atomicAdd(&cnt->errSymbol, 1ull);
    if (tooLong) atomicAdd(&cnt->errTooLong, 1ull);
}

// one thread per packed word: 21 codes -> 63 bits, first symbol in the top field
__global__ __launch_bounds__(256) void k_pack3(const u8 *__restrict__ T8, u64 n, u64 *__restrict__ text3, u64 nwords)
{
    for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (u64)gridDim.x * blockDim.x) {
        u64 p0 = w * BFQ_SYMS_PER_WORD;
        u64 v = 0;
        if (p0 + 24 <= n) {                                       // three unaligned 8-byte loads cover the 21 codes
            u64 x[3] = {*(const u64 *)(T8 + p0), *(const u64 *)(T8 + p0 + 8), *(const u64 *)(T8 + p0 + 16)};
#pragma unroll
            for (int j = 0; j < BFQ_SYMS_PER_WORD; j++) v = (v << 3) | ((x[j >> 3] >> (8 * (j & 7))) & 7ull);
        } else {
#pragma unroll 1
            for (int j = 0; j < BFQ_SYMS_PER_WORD; j++) {
                u64 p = p0 + j;
                u64 c = (p < n) ? (u64)T8[p] : 0ull;
                v = (v << 3) | c;
            }
        }
        text3[bfq_t3_at(w)] = v;
        const u64 j = w / BFQ_T3_PER_SEC, r = w - j * BFQ_T3_PER_SEC;
        if (r < 2 && j) text3[(j - 1) * 8 + 6 + r] = v;             // ... and as word 6 / 7 of the sector before
    }
}

// 4 consecutive suffixes per thread: 16-symbol key + payload (position, previous symbol, its quality) as
// 12-byte records; the previous symbols / qualities arrive by one 4-byte load each, the records leave as
// one 16-byte (w0) and two 16-byte (w12) stores
// STORE = false: only the digit counts (the sort's first scatter builds the records itself, straight from the text:
// k_radix_scatter<2>) -- no T8 / Q8 reads, no record stores.
template <bool STORE>
__global__ __launch_bounds__(256) void k_build_keys(const u8 *__restrict__ T8, const u8 *__restrict__ Q8,
                                                    const u64 *__restrict__ text3, u64 n, SortRec out, u32 *__restrict__ hist0,
                                                    u64 nblocks, u64 blockElems)
{
    // One workgroup per radix block (blockElems consecutive rows, a multiple of 1024), 1024 rows per sweep: the digit counts of
    // the sort's first pass come out on the way (per-wave LDS histograms), which saves that pass's 8 B/row histogram read.
    __shared__ u32 wh[4][256];
    const u32 w = threadIdx.x >> 6;
    // word index / symbol offset advance by 1024 rows per sweep: one division per block, not per row
    const u32 sw = 1024 / BFQ_SYMS_PER_WORD, so = 1024 - sw * BFQ_SYMS_PER_WORD;
    for (u64 hb = blockIdx.x; hb < nblocks; hb += gridDim.x) {
        for (int i = threadIdx.x; i < 4 * 256; i += 256) (&wh[0][0])[i] = 0;
        __syncthreads();
        const u64 bbase = hb * blockElems;
        u64 bend = bbase + blockElems;
        if (bend > n) bend = n;
        u64 p0 = bbase + (u64)threadIdx.x * 4;
        u64 w0 = p0 / BFQ_SYMS_PER_WORD;
        u32 o0 = (u32)(p0 - w0 * BFQ_SYMS_PER_WORD);
        for (; p0 < bend; p0 += 1024) {
            u32 c4 = 0, q4 = 0;                                       // codes / qualities of text positions p0-1 .. p0+2
            if (STORE) {
                if (p0 >= 1 && p0 + 3 <= n) { c4 = *(const u32 *)(T8 + p0 - 1); q4 = *(const u32 *)(Q8 + p0 - 1); }
                else for (int k = 0; k < 4; k++) { u64 t = p0 + k; if (t >= 1 && t - 1 < n) { c4 |= (u32)T8[t - 1] << (8 * k); q4 |= (u32)Q8[t - 1] << (8 * k); } }
            }
            const u64 *t3 = text3 + bfq_t3_at(w0);
            u64 t0 = t3[0], t1 = t3[1], t2 = t3[2];                   // 4 windows span at most 3 words
            u32 rw0[4];
            u64 rw12[4];
            u64 wd = w0;
            u32 o = o0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u64 a = (wd == w0) ? t0 : t1, bnext = (wd == w0) ? t1 : t2;
                u32 o3 = o * 3u;
                u64 hi = (a << o3) & BFQ_M63;
                u64 lo = o3 ? (bnext >> (63u - o3)) : 0ull;
                u64 sk = bfq_skey_of(bfq_mask_key(hi | lo));
                u32 pc = (c4 >> (8 * k)) & 0xFFu;
                u32 pq = pc ? (q4 >> (8 * k)) & 0xFFu : (u32)'#';
                u64 pay = bfq_pack_val(p0 + k, pc, pq);
                rw0[k] = bfq_rec_w0(sk);
                rw12[k] = ((u64)bfq_rec_w2(pay) << 32) | bfq_rec_w1(sk, pay);
                if (p0 + k < n) atomicAdd(&wh[w][(u32)sk & 255u], 1u);   // digit 0 of the LSD sort = low byte of the key
                if (++o == BFQ_SYMS_PER_WORD) { o = 0; wd++; }
            }
            if (!STORE) { }
            else if (p0 + 4 <= n) {
                *(uint4 *)(out.w0 + p0) = make_uint4(rw0[0], rw0[1], rw0[2], rw0[3]);
                *(ulonglong2 *)(out.w12 + p0) = make_ulonglong2(rw12[0], rw12[1]);
                *(ulonglong2 *)(out.w12 + p0 + 2) = make_ulonglong2(rw12[2], rw12[3]);
            } else {
                for (int k = 0; p0 + k < n; k++) { out.w0[p0 + k] = rw0[k]; out.w12[p0 + k] = rw12[k]; }
            }
            w0 += sw; o0 += so;
            if (o0 >= BFQ_SYMS_PER_WORD) { o0 -= BFQ_SYMS_PER_WORD; w0++; }
        }
        __syncthreads();
        hist0[(u64)threadIdx.x * nblocks + hb] = wh[0][threadIdx.x] + wh[1][threadIdx.x] + wh[2][threadIdx.x] + wh[3][threadIdx.x];
        __syncthreads();
    }
}

void bfq_build_text(bfq_ctx *c, const u8 *d_bases, const u8 *d_quals, const u64 *d_roff, u64 N, u64 n, u8 *T8,
                    u8 *Q8, u64 *text3, u64 nwords)
{
    if (N) {
        u64 blocks = bfq_grid(N, 16);                      // 16 lanes per read
        KLAUNCH(c, K_TEXT, 4.0 * (double)(n - N), k_text_from_reads, blocks, 256, d_bases, d_quals, d_roff, N, T8, Q8,
                c->d_cnt);
    }
    bfq_pack_text(c, T8, n, text3, nwords);
}


void bfq_pack_text(bfq_ctx *c, const u8 *T8, u64 n, u64 *text3, u64 nwords)
{
    const u64 nl = bfq_t3_logical(nwords);                      // text3 holds bfq_t3_alloc(nwords) words
    KLAUNCH(c, K_PACK, (double)n + 8.0 * (double)nwords, k_pack3, bfq_grid(nl, 256), 256, T8, n, text3, nl);
}

void bfq_build_keys(bfq_ctx *c, const u8 *T8, const u8 *Q8, const u64 *text3, u64 n, SortRec out, u32 *hist0)
{
    if (!n) return;
    const u64 be = bfq_radix_block_elems(n);
    const u64 nb = ceil_div(n, be);
    KLAUNCH(c, K_KEYS, 14.5 * (double)n, k_build_keys<true>, bfq_grid(nb, 1), 256, T8, Q8, text3, n, out, hist0, nb, be);
}
// only the first pass's digit counts (the records are made by that pass itself: bfq_radix_sort(.., fromText))
void bfq_key_hist(bfq_ctx *c, const u64 *text3, u64 n, u32 *hist0)
{
    if (!n) return;
    const u64 be = bfq_radix_block_elems(n);
    const u64 nb = ceil_div(n, be);
    KLAUNCH(c, K_KEYS, 0.4 * (double)n, k_build_keys<false>, bfq_grid(nb, 1), 256, (const u8 *)nullptr, (const u8 *)nullptr, text3, n, SortRec{nullptr, nullptr}, hist0, nb, be);
}
