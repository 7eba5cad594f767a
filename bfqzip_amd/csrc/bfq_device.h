// bfq_device.h -- device-only helpers (wave64 / workgroup primitives) for gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include "bfq_common.h"

#define BFQ_WAVE 64

__device__ __forceinline__ u32 bfq_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u64 bfq_lanemask_lt() { return (1ull << bfq_lane()) - 1ull; }

// 64-bit cross-lane moves built from the 32-bit primitives
__device__ __forceinline__ u64 bfq_readlane64(u64 v, int srcLane)   // srcLane wave-uniform
{
    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, srcLane);
    u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), srcLane);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 bfq_permute64(u64 v, int dstLane)     // push to a lane (dstLane a permutation)
{
    u32 lo = (u32)__builtin_amdgcn_ds_permute(dstLane << 2, (int)(u32)v);
    u32 hi = (u32)__builtin_amdgcn_ds_permute(dstLane << 2, (int)(u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// Lanes of ONE wavefront exchange data through LDS across this point: the scheduling barrier is fenced so that the
// compiler neither forwards a lane's own LDS value nor moves LDS accesses over it (free on a single wavefront).
__device__ __forceinline__ void bfq_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// whole-wavefront shifts by one lane (DPP wave_shl / wave_shr, gfx9): VALU moves, no LDS round trip
__device__ __forceinline__ u64 bfq_from_next_lane(u64 v)   // lane i <- lane i+1 (lane 63 <- 0)
{
    u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x130, 0xF, 0xF, true);
    u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x130, 0xF, 0xF, true);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 bfq_from_prev_lane(u64 v)   // lane i <- lane i-1 (lane 0 <- 0)
{
    u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x138, 0xF, 0xF, true);
    u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x138, 0xF, 0xF, true);
    return ((u64)hi << 32) | lo;
}

// inclusive wave scan (sum) of u64 / u32: DPP row shifts inside each row of 16 lanes (VALU moves,
// zero shifted in), then the totals of the rows before mine by three readlanes
#define BFQ_DPP_ROW_SHR(v, k) __builtin_amdgcn_update_dpp(0, (int)(v), 0x110 + (k), 0xF, 0xF, true)
__device__ __forceinline__ u32 bfq_wave_incscan32(u32 v)
{
    v += (u32)BFQ_DPP_ROW_SHR(v, 1);
    v += (u32)BFQ_DPP_ROW_SHR(v, 2);
    v += (u32)BFQ_DPP_ROW_SHR(v, 4);
    v += (u32)BFQ_DPP_ROW_SHR(v, 8);
    const u32 t0 = (u32)__builtin_amdgcn_readlane((int)v, 15), t1 = (u32)__builtin_amdgcn_readlane((int)v, 31),
              t2 = (u32)__builtin_amdgcn_readlane((int)v, 47);
    const u32 row = bfq_lane() >> 4;
    return v + (row >= 1 ? t0 : 0u) + (row >= 2 ? t1 : 0u) + (row >= 3 ? t2 : 0u);
}
__device__ __forceinline__ u64 bfq_row_shr64(u64 v, int k)   // k in {1,2,4,8}: compile-time after inlining
{
    u32 lo, hi;
    switch (k) {
    case 1: lo = (u32)BFQ_DPP_ROW_SHR((u32)v, 1); hi = (u32)BFQ_DPP_ROW_SHR((u32)(v >> 32), 1); break;
    case 2: lo = (u32)BFQ_DPP_ROW_SHR((u32)v, 2); hi = (u32)BFQ_DPP_ROW_SHR((u32)(v >> 32), 2); break;
    case 4: lo = (u32)BFQ_DPP_ROW_SHR((u32)v, 4); hi = (u32)BFQ_DPP_ROW_SHR((u32)(v >> 32), 4); break;
    default: lo = (u32)BFQ_DPP_ROW_SHR((u32)v, 8); hi = (u32)BFQ_DPP_ROW_SHR((u32)(v >> 32), 8); break;
    }
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 bfq_wave_incscan64(u64 v)
{
    v += bfq_row_shr64(v, 1);
    v += bfq_row_shr64(v, 2);
    v += bfq_row_shr64(v, 4);
    v += bfq_row_shr64(v, 8);
    const u64 t0 = bfq_readlane64(v, 15), t1 = bfq_readlane64(v, 31), t2 = bfq_readlane64(v, 47);
    const u32 row = bfq_lane() >> 4;
    return v + (row >= 1 ? t0 : 0ull) + (row >= 2 ? t1 : 0ull) + (row >= 3 ? t2 : 0ull);
}

// the same through the LDS crossbar (ds_bpermute): fewer live registers than the DPP sequence, for
// kernels at their VGPR limit (k_radix_scatter)
__device__ __forceinline__ u32 bfq_wave_incscan32_bp(u32 v)
{
    u32 lane = bfq_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = (u32)__builtin_amdgcn_ds_bpermute(((int)lane - d) << 2, (int)v);
        if ((int)lane >= d) v += t;
    }
    return v;
}

// exclusive scan over a 256-thread workgroup; sh must hold 4 entries; returns the
// exclusive prefix, *total = workgroup sum.  Contains two __syncthreads().
__device__ __forceinline__ u64 bfq_block_exscan64(u64 v, u64 *sh, u64 *total)
{
    u32 lane = bfq_lane(), w = threadIdx.x >> 6;
    u64 inc = bfq_wave_incscan64(v);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    u64 s0 = sh[0], s1 = sh[1], s2 = sh[2], s3 = sh[3];
    u64 base = (w > 0 ? s0 : 0) + (w > 1 ? s1 : 0) + (w > 2 ? s2 : 0);
    *total = s0 + s1 + s2 + s3;
    __syncthreads();
    return base + inc - v;
}
__device__ __forceinline__ u32 bfq_block_exscan32(u32 v, u32 *sh, u32 *total)
{
    u32 lane = bfq_lane(), w = threadIdx.x >> 6;
    u32 inc = bfq_wave_incscan32(v);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    u32 s0 = sh[0], s1 = sh[1], s2 = sh[2], s3 = sh[3];
    u32 base = (w > 0 ? s0 : 0) + (w > 1 ? s1 : 0) + (w > 2 ? s2 : 0);
    *total = s0 + s1 + s2 + s3;
    __syncthreads();
    return base + inc - v;
}
