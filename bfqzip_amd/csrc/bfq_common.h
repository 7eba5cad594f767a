// bfq_common.h -- helpers shared by host and device code (plain C++, no HIP types).
//
// Alphabet and order follow the reference's consumers: F array laid out
// # A C G N T (external/bwt2lcp/dna_bwt_n.hpp:46-61) and read i starting at BWT
// row i (src_int_mem/bfq_int.cpp:775-778), i.e. #_i < #_j (i<j) < A < C < G < N < T.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BFQ_HD __host__ __device__ __forceinline__
#else
#define BFQ_HD inline
#endif

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;

// ---- symbol codes: # 0, A 1, C 2, G 3, N 4, T 5 (order preserving) ----------
#define BFQ_CODE_INVALID 7
BFQ_HD u32 bfq_base_code(u8 c)
{
    return c == 'A' ? 1u : c == 'C' ? 2u : c == 'G' ? 3u : c == 'N' ? 4u : c == 'T' ? 5u
                                                                              : (u32)BFQ_CODE_INVALID;
}
BFQ_HD u8 bfq_code_sym(u32 code)   // code 0..5 -> ASCII "#ACGNT" (byte LUT in a u64)
{
    return (u8)((0x0000544E47434123ull >> (8u * code)) & 0xFFu);
}

// ---- packed text: 21 symbols of 3 bits per 64-bit word, symbol j of a word in
// bits [3*(20-j), 3*(20-j)+2]; bit 63 is always 0.  A "key" is the 21-symbol
// window starting at any text position, with everything from the first
// terminator on cleared, so that keys compare like the suffixes' first 21
// symbols and equal keys holding a terminator are identical suffixes.
#define BFQ_SYMS_PER_WORD 21
#define BFQ_M63 0x7FFFFFFFFFFFFFFFull
#define BFQ_LOW3 0x1249249249249249ull   // bit 0 of each of the 21 fields

BFQ_HD int bfq_clz64(u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return x ? __builtin_clzll(x) : 64;
#endif
}
BFQ_HD int bfq_popc64(u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

// mask of the low bit of every zero field of a 63-bit window
BFQ_HD u64 bfq_zero_fields(u64 k)
{
    u64 t = k | (k >> 1) | (k >> 2);
    return ~t & BFQ_LOW3;
}
// clear everything after the first terminator (zero field)
BFQ_HD u64 bfq_mask_key(u64 k)
{
    u64 z = bfq_zero_fields(k);
    if (z == 0) return k;
    int b = 63 - bfq_clz64(z);            // low bit of the first zero field (0,3,..,60)
    return k & ~((1ull << (b + 3)) - 1ull);
}
BFQ_HD bool bfq_key_has_term(u64 maskedKey) { return bfq_zero_fields(maskedKey) != 0; }
// number of symbols before the first terminator (21 if none)
BFQ_HD int bfq_key_tpos(u64 maskedKey)
{
    u64 z = bfq_zero_fields(maskedKey);
    if (z == 0) return BFQ_SYMS_PER_WORD;
    int b = 63 - bfq_clz64(z);
    return (60 - b) / 3;
}
// common prefix (in symbols) of two masked windows; terminators never match
BFQ_HD int bfq_key_lcp(u64 a, u64 b)
{
    u64 x = a ^ b;
    if (x == 0) return bfq_key_tpos(a);
    return (bfq_clz64(x) - 1) / 3;
}
// The packed text in memory: 64-byte sectors of 8 words, sector j holding the logical words [6 j, 6 j + 8) -- its last two
// words repeat the first two of sector j + 1.  Any THREE consecutive logical words (what a refinement round, a key window
// or a record builder reads) therefore lie inside ONE sector: a fetch never straddles two (a quarter of the 24-byte reads of
// a plain array did, and the refinement is bound by the number of sectors it waits for).  Costs a third more text (0.51
// instead of 0.38 bytes per symbol) and one division by 6 per fetch.
#define BFQ_T3_PER_SEC 6
BFQ_HD u64 bfq_t3_at(u64 w) { const u64 j = w / BFQ_T3_PER_SEC; return j * 8 + (w - j * BFQ_T3_PER_SEC); }   // where logical word w (and w + 1, w + 2) is read
BFQ_HD u64 bfq_t3_logical(u64 nwords) { return (nwords / BFQ_T3_PER_SEC + 1) * BFQ_T3_PER_SEC + 2; }         // logical words to pack (the tail repeats are zero-filled that way)
BFQ_HD u64 bfq_t3_alloc(u64 nwords) { return (nwords / BFQ_T3_PER_SEC + 2) * 8; }                            // u64s to allocate for nwords logical words
// raw 21-symbol window at text position p (the text is padded with >= 2 zero words)
BFQ_HD u64 bfq_window(const u64 *text3, u64 p)
{
    u64 w = p / BFQ_SYMS_PER_WORD;
    u32 o = (u32)(p - w * BFQ_SYMS_PER_WORD) * 3u;
    const u64 *t = text3 + bfq_t3_at(w);
    u64 hi = (t[0] << o) & BFQ_M63;
    u64 lo = o ? (t[1] >> (63u - o)) : 0ull;
    return hi | lo;
}
BFQ_HD u64 bfq_key_at(const u64 *text3, u64 p) { return bfq_mask_key(bfq_window(text3, p)); }

// ---- sort record of one suffix: three 32-bit words (12 bytes per row), w0 in one array, (w1, w2) in another
//   skey    = the suffix's first 16 symbols as key40 | L << 40
//             key40: two symbols (c0, c1) -> one 5-bit group, 0 for "#", else 6*c0 + c1 - 5 (1..30).  After a terminator
//             every symbol is 0, so the map is one-to-one and keeps the order: 8 groups = 40 bits = five 8-bit radix
//             digits instead of the six a 3-bit-per-symbol key needs.
//             L: symbols before the first terminator among the 16 (16: none) -- a function of key40, carried along
//             so that "complete suffix" and the common prefix of equal keys need no decoding.
//   payload = text position (37 bits) | preceding symbol code << 37 | its quality << 40   (48 bits)
//   w0 = key40 >> 8 ;  w1 = (key40 & 0xFF) << 24 | L << 16 | payload >> 32 ;  w2 = payload & 0xFFFFFFFF
// 8-bit radix digits of key40: digit 0 lives in w1 (bits 24..31), digits 1..4 in w0.
#define BFQ_KEY_SYMS 16
#define BFQ_KEY_PASSES 5
#define BFQ_POS_BITS 37
#define BFQ_POS_MASK ((1ull << BFQ_POS_BITS) - 1ull)
BFQ_HD u64 bfq_key40_of_key48(u64 X)                 // X: 16 fields of 3 bits, first symbol on top, zero after a terminator
{
    const u64 M7 = 0x1C71C71C71C7ull, L1 = 0x041041041041ull;
    u64 A = (X >> 3) & M7, B = X & M7;               // c0 / c1 of the 8 pairs, one pair per 6-bit lane
    u64 S = (A << 2) + (A << 1) + B;                 // 6*c0 + c1 (<= 35: stays inside the lane)
    u64 nz = (A | (A >> 1) | (A >> 2)) & L1;
    S -= (nz << 2) + nz;                             // - 5 where c0 != 0
    u64 v = (S & 0x03F03F03F03Full) | ((S & 0xFC0FC0FC0FC0ull) >> 1);      // 8 x 6 -> 4 x 10 -> 2 x 20 -> 40 bits
    v = (v & 0x000FFF000FFFull) | ((v & 0xFFF000FFF000ull) >> 2);
    return (v & 0xFFFFFull) | ((v >> 24) << 20);
}
BFQ_HD u64 bfq_skey_of(u64 maskedKey63)
{
    int t = bfq_key_tpos(maskedKey63);
    return bfq_key40_of_key48(maskedKey63 >> 15) | ((u64)(t < BFQ_KEY_SYMS ? t : BFQ_KEY_SYMS) << 40);
}
BFQ_HD u64 bfq_pack_val(u64 pos, u32 prevCode, u32 prevQual)
{
    return pos | ((u64)prevCode << 37) | ((u64)(prevQual & 0xFFu) << 40);
}
BFQ_HD u64 bfq_val_pos(u64 v) { return v & BFQ_POS_MASK; }
BFQ_HD u32 bfq_val_code(u64 v) { return (u32)(v >> 37) & 7u; }
BFQ_HD u32 bfq_val_qual(u64 v) { return (u32)(v >> 40) & 0xFFu; }
// (any integer below 2^40 can stand in for an skey: the segment-id sorts of k_bigseg.hip)
BFQ_HD u32 bfq_rec_w0(u64 skey) { return (u32)(skey >> 8); }
BFQ_HD u32 bfq_rec_w1(u64 skey, u64 pay) { return ((u32)(skey & 0xFFu) << 24) | (((u32)(skey >> 40) & 31u) << 16) | (u32)(pay >> 32); }
BFQ_HD u32 bfq_rec_w2(u64 pay) { return (u32)pay; }
BFQ_HD u64 bfq_rec_skey(u32 w0, u32 w1) { return ((u64)w0 << 8) | (u64)(w1 >> 24) | ((u64)((w1 >> 16) & 31u) << 40); }
BFQ_HD u64 bfq_rec_pay(u32 w1, u32 w2) { return ((u64)(w1 & 0xFFFFu) << 32) | (u64)w2; }
// the key holds a terminator among its 16 symbols = it is a complete suffix
BFQ_HD bool bfq_skey_has_term(u64 k) { return (u32)(k >> 40) < (u32)BFQ_KEY_SYMS; }
// common prefix (symbols) of two skeys; terminators never match
BFQ_HD int bfq_skey_lcp(u64 a, u64 b)
{
    u64 x = (a ^ b) & 0xFFFFFFFFFFull;
    if (x == 0) return (int)(a >> 40);
    int gi = ((bfq_clz64(x) - 24) * 13) >> 6;        // first group that differs (/5 for 0..39)
    int sh = 35 - 5 * gi;
    u32 ga = (u32)(a >> sh) & 31u, gb = (u32)(b >> sh) & 31u;
    u32 ca = ((ga + 5u) * 43u) >> 8, cb = ((gb + 5u) * 43u) >> 8;   // the group's first symbol: (g + 5) / 6
    return 2 * gi + (ca == cb ? 1 : 0);
}

// ---- Illumina 8-level binning, ASCII in/out (bfq_int.cpp:307-319) ------------
BFQ_HD u32 bfq_bin8(u32 asciiQ)
{
    int q = (int)(signed char)asciiQ - 33;
    if (q >= 40) q = 40; else if (q >= 35) q = 37; else if (q >= 30) q = 33;
    else if (q >= 25) q = 27; else if (q >= 20) q = 22; else if (q >= 10) q = 15;
    else if (q >= 2) q = 6;
    return (u32)(q + 33) & 0xFFu;
}

// ---- counter-based hash for the synthetic generator --------------------------
BFQ_HD u64 bfq_mix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
BFQ_HD u64 bfq_hash2(u64 seed, u64 tag, u64 x) { return bfq_mix64(bfq_mix64(seed ^ (tag * 0xD6E8FEB86659FD93ull)) + x); }
