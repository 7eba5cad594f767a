// bfq_rec.h -- device accessors of the 12-byte sort records (layout: bfq_common.h)
#pragma once
#include "bfq_internal.h"

// row r starts a segment: its 16-symbol key differs from the previous row's, or holds a terminator
__device__ __forceinline__ bool seg_head(u64 kp, u64 k) { return (k != kp) || bfq_skey_has_term(k); }
__device__ __forceinline__ u64 rec_key(const SortRec &r, u64 i) { return bfq_rec_skey(r.w0[i], (u32)r.w12[i]); }
__device__ __forceinline__ u64 rec_pay(const SortRec &r, u64 i) { u64 x = r.w12[i]; return bfq_rec_pay((u32)x, (u32)(x >> 32)); }
// rows of a segment share the key, so its low half can be rewritten from any of them
__device__ __forceinline__ void rec_set_pay(const SortRec &r, u64 i, u64 pay)
{
    u32 w1 = ((u32)r.w12[i] & 0xFFFF0000u) | (u32)(pay >> 32);
    r.w12[i] = ((u64)(u32)pay << 32) | w1;
}
