// Host-side unit test of bfq_common.h (the helpers the kernels share with the host):
// key windows over the 3-bit packed text, terminator masking, LCP of keys, the 40-bit sort key, sort-record packing,
// Illumina binning.  Compared against byte-wise definitions.  Exit code 0 = all good.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../bfqzip_amd/csrc/bfq_common.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main()
{
    // a text of codes: reads separated by terminators (0)
    std::vector<u8> T;
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int r = 0; r < 200; r++) {
        int len = rnd() % 60;
        for (int k = 0; k < len; k++) T.push_back(1 + rnd() % 5);
        T.push_back(0);
    }
    size_t n = T.size();
    // the packed text as the kernels lay it out (k_pack3): sectors of 8 words holding the logical words [6 j, 6 j + 8)
    const size_t nwords = n / BFQ_SYMS_PER_WORD + 3;
    std::vector<u64> text3(bfq_t3_alloc(nwords), 0xDEADBEEFDEADBEEFull);      // what is never written must never be read
    std::vector<u64> plain(bfq_t3_logical(nwords), 0);
    for (size_t w = 0; w < plain.size(); w++) {
        u64 v = 0;
        for (int j = 0; j < BFQ_SYMS_PER_WORD; j++) { size_t p = w * BFQ_SYMS_PER_WORD + j; v = (v << 3) | (p < n ? T[p] : 0); }
        plain[w] = v;
        text3[bfq_t3_at(w)] = v;
        const size_t js = w / BFQ_T3_PER_SEC, r = w - js * BFQ_T3_PER_SEC;
        if (r < 2 && js) text3[(js - 1) * 8 + 6 + r] = v;
    }
    for (size_t w = 0; w + 2 < plain.size() && w < nwords; w++)                // any three consecutive words: one sector, in order
        for (int k = 0; k < 3; k++) {
            CHECK(text3[bfq_t3_at(w) + k] == plain[w + k]);
            CHECK((bfq_t3_at(w) + k) / 8 == bfq_t3_at(w) / 8);
        }
    auto sym = [&](u64 key, int j) { return (int)((key >> (3 * (20 - j))) & 7); };
    for (size_t p = 0; p < n; p++) {
        u64 key = bfq_key_at(text3.data(), p);
        bool dead = false;
        int tpos = BFQ_SYMS_PER_WORD;
        for (int j = 0; j < BFQ_SYMS_PER_WORD; j++) {
            int want = (!dead && p + j < n) ? T[p + j] : 0;
            if (!dead && want == 0) { dead = true; tpos = j; }
            CHECK(sym(key, j) == (dead ? 0 : want));
        }
        CHECK(bfq_key_has_term(key) == (tpos < BFQ_SYMS_PER_WORD));
        CHECK(bfq_key_tpos(key) == tpos);
        CHECK((key >> 63) == 0);
        // sort key = first 16 symbols in 40 bits + the terminator's place
        u64 sk = bfq_skey_of(key);
        CHECK((sk >> 45) == 0);
        CHECK(bfq_skey_has_term(sk) == (tpos < BFQ_KEY_SYMS));
        CHECK((int)(sk >> 40) == (tpos < BFQ_KEY_SYMS ? tpos : BFQ_KEY_SYMS));
        u64 pay = bfq_pack_val(p, p ? T[p - 1] : 0, 33 + (u32)(p % 90));
        u32 w0 = bfq_rec_w0(sk), w1 = bfq_rec_w1(sk, pay), w2 = bfq_rec_w2(pay);
        CHECK(bfq_rec_skey(w0, w1) == sk);
        CHECK(bfq_rec_pay(w1, w2) == pay);
        CHECK(bfq_val_pos(pay) == p && bfq_val_code(pay) == (p ? T[p - 1] : 0u) && bfq_val_qual(pay) == 33 + (u32)(p % 90));
    }
    // LCP of two keys = common prefix on bases only (terminators never match), order = suffix order
    for (int it = 0; it < 20000; it++) {
        size_t p = rnd() % n, q = rnd() % n;
        u64 a = bfq_key_at(text3.data(), p), b = bfq_key_at(text3.data(), q);
        int l = 0;
        while (l < BFQ_SYMS_PER_WORD && T[p + l] && p + l < n && q + l < n && T[p + l] == T[q + l]) l++;
        CHECK(bfq_key_lcp(a, b) == l);
        int l16 = l < BFQ_KEY_SYMS ? l : BFQ_KEY_SYMS;
        const u64 sa = bfq_skey_of(a), sb = bfq_skey_of(b);
        CHECK(bfq_skey_lcp(sa, sb) == l16);
        // the 40-bit key keeps the order and the equality of the first 16 symbols (key >> 15: 3 bits per symbol)
        const u64 ka = a >> 15, kb = b >> 15, M40 = (1ull << 40) - 1ull;
        CHECK((ka < kb) == ((sa & M40) < (sb & M40)));
        CHECK((ka == kb) == (sa == sb));
        // lexicographic order of the first 21 symbols (# smallest)
        int c = 0;
        for (int j = 0; j < BFQ_SYMS_PER_WORD && !c; j++) {
            int x = (p + j < n) ? T[p + j] : 0, y = (q + j < n) ? T[q + j] : 0;
            if (x != y) c = x < y ? -1 : 1;
            if (x == 0 || y == 0) break;
        }
        CHECK((a < b) == (c < 0) || c == 0);
    }
    // pairs of 21-symbol windows with a chosen common prefix (random text rarely shares more than a few symbols)
    for (int it = 0; it < 200000; it++) {
        int sa_[21], sb_[21];
        const int share = rnd() % 22, la = rnd() % 22, lb = rnd() % 22;       // la / lb: place of the terminator (21: none)
        for (int j = 0; j < 21; j++) { sa_[j] = 1 + rnd() % 5; sb_[j] = (j < share) ? sa_[j] : 1 + rnd() % 5; }
        u64 a = 0, b = 0;
        for (int j = 0; j < 21; j++) { a = (a << 3) | (u64)(j < la ? sa_[j] : 0); b = (b << 3) | (u64)(j < lb ? sb_[j] : 0); }
        int l = 0;
        while (l < 16 && l < la && l < lb && sa_[l] == sb_[l]) l++;
        const u64 ka = a >> 15, kb = b >> 15, M40 = (1ull << 40) - 1ull;
        const u64 sa = bfq_skey_of(a), sb = bfq_skey_of(b);
        CHECK(bfq_skey_lcp(sa, sb) == l);
        CHECK((ka < kb) == ((sa & M40) < (sb & M40)));
        CHECK((ka == kb) == (sa == sb));
    }
    for (int q = 0; q < 128; q++) {   // bfq_int.cpp:307-319
        int v = q - 33, e = v;
        if (v >= 40) e = 40; else if (v >= 35) e = 37; else if (v >= 30) e = 33; else if (v >= 25) e = 27;
        else if (v >= 20) e = 22; else if (v >= 10) e = 15; else if (v >= 2) e = 6;
        CHECK(bfq_bin8(q) == (u32)((e + 33) & 0xFF));
    }
    CHECK(bfq_code_sym(0) == '#' && bfq_code_sym(1) == 'A' && bfq_code_sym(2) == 'C' && bfq_code_sym(3) == 'G' &&
          bfq_code_sym(4) == 'N' && bfq_code_sym(5) == 'T');
    for (int c = 0; c < 256; c++) {
        u32 k = bfq_base_code((u8)c);
        CHECK((k != BFQ_CODE_INVALID) == (c == 'A' || c == 'C' || c == 'G' || c == 'N' || c == 'T'));
    }
    printf(fails ? "%d failures\n" : "ok\n", fails);
    return fails ? 1 : 0;
}
