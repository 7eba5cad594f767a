// k_fastq.hip -- FASTQ text <-> read arrays on the GPU.
//
// The reference's drivers pull the streams apart with `sed -n 1~4p / 2~4p / 4~4p`
// (BFQzip.py:192-251) and bfq_int writes records with four fwrite()s per read
// (bfq_int.cpp:797-810).  Here the 4-line records are indexed and gathered / written by
// streaming kernels, so the front-ends hand whole files to the GPU:
//   k_nl_count / k_nl_write : positions of all line ends (ordered compaction)
//   k_fq_records            : per record: header span, sequence / quality start, length
//                             (CR before LF stripped; len(DNA) != len(QS) is an error, checkFASTQ.py:18-32)
//   k_fq_gather             : lines 2 and 4 -> bases / quals back to back (one wave per read)
//   k_fq_format             : header line (verbatim, or "@"), bases, "+", quals (one wave per read)
//   k_fq_hdr_gather         : the header stream of BFQzip.py --m3 (OUT.h): every read's header as one line
//                             (OUT.fq.dna / OUT.fq.qs are written by the inversion kernel itself)
#include "bfq_internal.h"
#include "bfq_device.h"

#define NL_CHUNK 4096                                  // bytes per workgroup iteration (16 per thread)

__device__ __forceinline__ u32 nl_mask16(const u8 *__restrict__ buf, u64 pos, u64 len)
{
    u32 m = 0;
    if (pos + 16 <= len) {
        uint4 v = *(const uint4 *)(buf + pos);         // buf is 16-byte aligned, pos a multiple of 16
        u32 wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int b = 0; b < 4; b++) m |= (((wds[k] >> (8 * b)) & 0xFFu) == 10u ? 1u : 0u) << (4 * k + b);
    } else {
        for (int k = 0; k < 16; k++)
            if (pos + k < len && buf[pos + k] == 10) m |= 1u << k;
    }
    return m;
}

__global__ __launch_bounds__(256) void k_nl_count(const u8 *__restrict__ buf, u64 len, u32 *__restrict__ counts, u64 nchunks)
{
    __shared__ u32 sh[4];
    for (u64 ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        u32 c = __popc(nl_mask16(buf, ch * NL_CHUNK + (u64)threadIdx.x * 16, len));
        u32 tot;
        bfq_block_exscan32(c, sh, &tot);
        if (threadIdx.x == 0) counts[ch] = tot;
    }
}
__global__ __launch_bounds__(256) void k_nl_write(const u8 *__restrict__ buf, u64 len, const u64 *__restrict__ chunkBase,
                                                  u64 *__restrict__ lineEnd, u64 nchunks)
{
    __shared__ u32 sh[4];
    for (u64 ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        u64 pos = ch * NL_CHUNK + (u64)threadIdx.x * 16;
        u32 m = nl_mask16(buf, pos, len);
        u32 tot;
        u32 ex = bfq_block_exscan32(__popc(m), sh, &tot);
        u64 o = chunkBase[ch] + ex;
        while (m) { int k = __builtin_ctz(m); lineEnd[o++] = pos + k; m &= m - 1; }
    }
}

struct FqRec { u64 hdrStart, seqStart, qualStart; u32 hdrLen, len; };

__global__ __launch_bounds__(256) void k_fq_records(const u8 *__restrict__ buf, const u64 *__restrict__ lineEnd, u64 N, FqRec *__restrict__ rec, u32 *__restrict__ lens,
                                                    DevCounters *cnt)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        u64 e[4], s[4];
        for (int k = 0; k < 4; k++) {
            u64 li = 4 * i + k;
            s[k] = li ? lineEnd[li - 1] + 1 : 0;
            e[k] = lineEnd[li];
            if (k && e[k] > s[k] && buf[e[k] - 1] == 13) e[k]--;    // CRLF (header lines are passed through verbatim)
        }
        FqRec r;
        r.hdrStart = s[0]; r.hdrLen = (u32)(e[0] - s[0]);
        r.seqStart = s[1]; r.qualStart = s[3];
        u64 l = e[1] - s[1];
        if (e[3] - s[3] != l) atomicAdd(&cnt->errFastq, 1ull);
        if (l > BFQ_MAX_READ_LEN) { atomicAdd(&cnt->errTooLong, 1ull); l = 0; }
        r.len = (u32)l;
        rec[i] = r;
        lens[i] = (u32)l;
    }
}

// copy len bytes with unaligned 8-byte accesses, `sub` of `nsub` lanes working together; the last word overlaps the one
// before (no byte tail); len < 8: lane 0 alone, byte by byte
__device__ __forceinline__ void copy_bytes(u8 *__restrict__ dst, const u8 *__restrict__ src, u64 len, u32 sub, u32 nsub)
{
    if (len >= 8) {
        const u64 nw = (len + 7) >> 3;
        for (u64 j = sub; j < nw; j += nsub) {
            const u64 k = (8 * j + 8 <= len) ? 8 * j : len - 8;
            *(u64 *)(dst + k) = *(const u64 *)(src + k);
        }
    } else if (sub == 0) {
        for (u64 k = 0; k < len; k++) dst[k] = src[k];
    }
}

// 16 lanes per read, 8 bytes per lane and step
__global__ __launch_bounds__(256) void k_fq_gather(const u8 *__restrict__ buf, const FqRec *__restrict__ rec,
                                                   const u64 *__restrict__ roff, u64 N, u8 *__restrict__ bases,
                                                   u8 *__restrict__ quals)
{
    const u32 sub = threadIdx.x & 15u;
    const u64 ngrp = ((u64)gridDim.x * blockDim.x) >> 4;
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < N; i += ngrp) {
        const FqRec r = rec[i];
        const u64 o = roff[i];
        copy_bytes(bases + o, buf + r.seqStart, r.len, sub, 16);
        copy_bytes(quals + o, buf + r.qualStart, r.len, sub, 16);
    }
}

// record i: header (hLen[i] bytes at hdr + hStart[i], or "@" when hdr == nullptr) \n bases \n + \n quals \n
__global__ __launch_bounds__(256) void k_fq_hdr_from_lines(const u64 *__restrict__ hdrEnd, u64 N, u64 *__restrict__ hStart,
                                                           u32 *__restrict__ hLen)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        u64 s0 = i ? hdrEnd[i - 1] + 1 : 0;
        hStart[i] = s0; hLen[i] = (u32)(hdrEnd[i] - s0);
    }
}
__global__ __launch_bounds__(256) void k_fq_hdr_from_recs(const FqRec *__restrict__ rec, u64 N, u64 *__restrict__ hStart,
                                                          u32 *__restrict__ hLen)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        hStart[i] = rec[i].hdrStart; hLen[i] = rec[i].hdrLen;
    }
}
__global__ __launch_bounds__(256) void k_fq_recsize(const u64 *__restrict__ roff, const u32 *__restrict__ hLen, u64 N,
                                                    u32 *__restrict__ sizes)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        u64 L = roff[i + 1] - roff[i];
        sizes[i] = (u32)((hLen ? hLen[i] : 1u) + 2 * L + 5);
    }
}
__global__ __launch_bounds__(256) void k_fq_format(const u8 *__restrict__ bases, const u8 *__restrict__ quals,
                                                   const u64 *__restrict__ roff, const u8 *__restrict__ hdr,
                                                   const u64 *__restrict__ hStart, const u32 *__restrict__ hLen,
                                                   const u64 *__restrict__ recOff, u64 N, u8 *__restrict__ out, int lines)
{
    const u32 sub = threadIdx.x & 15u;                           // 16 lanes per record, 8 bytes per lane and step
    const u64 ngrp = ((u64)gridDim.x * blockDim.x) >> 4;
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < N; i += ngrp) {
        u64 b = roff[i], L = roff[i + 1] - b, o = recOff[i];
        if (lines) b += i;                                       // bases / quals given as line streams (read i at roff[i] + i)
        const u64 hl = hdr ? hLen[i] : 1;
        if (hdr) copy_bytes(out + o, hdr + hStart[i], hl, sub, 16);
        else if (sub == 0) out[o] = (u8)'@';
        o += hl;
        if (sub == 0) { out[o] = 10; out[o + 1 + L] = 10; out[o + 2 + L] = (u8)'+'; out[o + 3 + L] = 10; out[o + 4 + 2 * L] = 10; }
        copy_bytes(out + o + 1, bases + b, L, sub, 16);
        copy_bytes(out + o + 4 + L, quals + b, L, sub, 16);
    }
}

__global__ __launch_bounds__(256) void k_fq_hdrsize(const FqRec *__restrict__ rec, u64 N, u32 *__restrict__ sizes)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) sizes[i] = rec[i].hdrLen + 1;
}
__global__ __launch_bounds__(256) void k_fq_hdr_gather(const u8 *__restrict__ buf, const FqRec *__restrict__ rec,
                                                       const u64 *__restrict__ hOff, u64 N, u8 *__restrict__ out)
{
    u32 lane = bfq_lane();
    u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < N; i += nwaves) {
        u64 s = rec[i].hdrStart, hl = rec[i].hdrLen, o = hOff[i];
        for (u64 k = lane; k < hl; k += 64) out[o + k] = buf[s + k];
        if (lane == 0) out[o + hl] = 10;
    }
}

// positions of the line ends of a text resident on the device; a last line without newline gets a
// virtual end at `len`.  *nlines = number of lines.
u64 *bfq_line_index(bfq_ctx *c, const u8 *d_buf, u64 len, u64 *nlines)
{
    u64 nchunks = ceil_div(len ? len : 1, NL_CHUNK);
    u32 *counts = c->alloc<u32>(nchunks);
    u64 *bases = c->alloc<u64>(nchunks);
    u64 *d_total = c->alloc<u64>(1);
    KLAUNCH(c, K_FASTQ, (double)len, k_nl_count, bfq_grid(nchunks, 1), 256, d_buf, len, counts, nchunks);
    bfq_exscan_u32(c, counts, bases, nchunks, d_total);
    u64 nl = 0;
    HIP_CHECK(hipMemcpyAsync(&nl, d_total, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    u8 last = 10;
    if (len) HIP_CHECK(hipMemcpyAsync(&last, d_buf + len - 1, 1, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    u64 *lineEnd = c->alloc<u64>(nl + 2);
    if (nl) KLAUNCH(c, K_FASTQ, (double)len + 8.0 * (double)nl, k_nl_write, bfq_grid(nchunks, 1), 256, d_buf, len,
                    (const u64 *)bases, lineEnd, nchunks);
    if (len && last != 10) {
        HIP_CHECK(hipMemcpyAsync(lineEnd + nl, &len, sizeof(u64), hipMemcpyHostToDevice, c->stream));
        c->sync();                                              // &len is a stack variable
        nl++;
    }
    *nlines = nl;
    return lineEnd;
}

u64 bfq_fastq_count_lines(bfq_ctx *c, const u8 *d_buf, u64 len)
{
    size_t m = c->mark();
    u64 nchunks = ceil_div(len ? len : 1, NL_CHUNK);
    u32 *counts = c->alloc<u32>(nchunks);
    u64 *bases = c->alloc<u64>(nchunks);
    u64 *d_total = c->alloc<u64>(1);
    KLAUNCH(c, K_FASTQ, (double)len, k_nl_count, bfq_grid(nchunks, 1), 256, d_buf, len, counts, nchunks);
    bfq_exscan_u32(c, counts, bases, nchunks, d_total);
    u64 nl = 0;
    HIP_CHECK(hipMemcpyAsync(&nl, d_total, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    c->release(m);
    return nl + 1;
}

void bfq_fastq_parse(bfq_ctx *c, const u8 *d_fastq, u64 len, DevFastq *fq)
{
    u64 nlines = 0;
    u64 *lineEnd = bfq_line_index(c, d_fastq, len, &nlines);
    if (nlines % 4) throw BfqError{BFQ_E_ARG, "FASTQ: number of lines is not a multiple of 4"};
    u64 N = nlines / 4;
    fq->N = N;
    fq->rec = c->alloc<FqRec>(N + 1);
    fq->roff = c->alloc<u64>(N + 2);
    u32 *lens = c->alloc<u32>(N + 1);
    if (N)
        KLAUNCH(c, K_FASTQ, 40.0 * (double)N, k_fq_records, bfq_grid(N, 256), 256, d_fastq, (const u64 *)lineEnd, N,
                (FqRec *)fq->rec, lens, c->d_cnt);
    bfq_exscan_u32(c, lens, fq->roff, N, fq->roff + N);
    HIP_CHECK(hipMemcpyAsync(&fq->total, fq->roff + N, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    c->fetchCounters();
    if (c->h_cnt.errFastq) throw BfqError{BFQ_E_ARG, "FASTQ: len(DNA) != len(QS) in a record"};
    if (c->h_cnt.errTooLong) throw BfqError{BFQ_E_TOO_LONG, "read longer than BFQ_MAX_READ_LEN"};
    fq->bases = c->alloc<u8>(fq->total + 64);
    fq->quals = c->alloc<u8>(fq->total + 64);
    if (N) {
        KLAUNCH(c, K_FASTQ, 4.0 * (double)fq->total, k_fq_gather, bfq_grid(N, 16), 256, d_fastq, (const FqRec *)fq->rec,
                (const u64 *)fq->roff, N, fq->bases, fq->quals);
    }
    fq->lineEnd = lineEnd;
}

// Headers: mode 0 = "@"; 1 = d_hdr is a text of header lines (bfq_int -H); 2 = d_hdr is the FASTQ
// text parsed into `fq` (its records' own header lines).  Returns the formatted length, text in *d_out (arena).
u64 bfq_fastq_format(bfq_ctx *c, const u8 *d_bases, const u8 *d_quals, const u64 *d_roff, u64 N, int mode, const u8 *d_hdr,
                     u64 hdrLen, const DevFastq *fq, u8 **d_out, u64 **recOffOut, bool lines)
{
    u64 *hStart = nullptr;
    u32 *hLen = nullptr;
    if (mode) {
        hStart = c->alloc<u64>(N + 1);
        hLen = c->alloc<u32>(N + 1);
        if (mode == 1) {
            u64 nl = 0;
            const u64 *hdrEnd = bfq_line_index(c, d_hdr, hdrLen, &nl);
            if (nl < N) throw BfqError{BFQ_E_ARG, "header file has fewer lines than there are reads"};
            if (N) KLAUNCH(c, K_FASTQ, 20.0 * (double)N, k_fq_hdr_from_lines, bfq_grid(N, 256), 256, hdrEnd, N, hStart, hLen);
        } else if (N) {
            KLAUNCH(c, K_FASTQ, 44.0 * (double)N, k_fq_hdr_from_recs, bfq_grid(N, 256), 256, (const FqRec *)fq->rec, N, hStart, hLen);
        }
    }
    u32 *sizes = c->alloc<u32>(N + 1);
    u64 *recOff = c->alloc<u64>(N + 2);
    if (N) KLAUNCH(c, K_FASTQ, 12.0 * (double)N, k_fq_recsize, bfq_grid(N, 256), 256, d_roff, (const u32 *)hLen, N, sizes);
    bfq_exscan_u32(c, sizes, recOff, N, recOff + N);
    u64 outLen = 0;
    HIP_CHECK(hipMemcpyAsync(&outLen, recOff + N, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    u8 *out = c->alloc<u8>(outLen + 64);
    if (N) {
        KLAUNCH(c, K_FASTQ, 2.0 * (double)outLen, k_fq_format, bfq_grid(N, 16), 256, d_bases, d_quals, d_roff,
                mode ? d_hdr : (const u8 *)nullptr, (const u64 *)hStart, (const u32 *)hLen, (const u64 *)recOff, N, out, lines ? 1 : 0);
    }
    *d_out = out;
    if (recOffOut) *recOffOut = recOff;
    return outLen;
}

// The header stream of BFQzip.py --m3 (OUT.h = `sed -n 1~4p in.fastq`): the header lines of the parsed FASTQ `fq`, one
// per line; *hdrLen bytes in *d_hdr, line offsets in *hOffOut.  (The OUT.fq.dna / OUT.fq.qs streams are written by the
// inversion itself, k_invert<.., 1>.)
void bfq_fastq_hdr_stream(bfq_ctx *c, u64 N, const u8 *d_fastq, const DevFastq *fq, u8 **d_hdr, u64 *hdrLen, u64 **hOffOut)
{
    u64 waves = N < (1u << 18) ? N : (1u << 18);
    u32 *sizes = c->alloc<u32>(N + 1);
    u64 *hOff = c->alloc<u64>(N + 2);
    if (N) KLAUNCH(c, K_FASTQ, 28.0 * (double)N, k_fq_hdrsize, bfq_grid(N, 256), 256, (const FqRec *)fq->rec, N, sizes);
    bfq_exscan_u32(c, sizes, hOff, N, hOff + N);
    u64 hl = 0;
    HIP_CHECK(hipMemcpyAsync(&hl, hOff + N, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    u8 *hdr = c->alloc<u8>(hl + 64);
    if (N) KLAUNCH(c, K_FASTQ, 2.0 * (double)hl, k_fq_hdr_gather, ceil_div(waves, 4), 256, d_fastq, (const FqRec *)fq->rec, (const u64 *)hOff, N, hdr);
    *d_hdr = hdr; *hdrLen = hl;
    if (hOffOut) *hOffOut = hOff;
}

// ---- parts of a job (bfq_fastq_run_job): index of the first record of every part = number of records whose header
// line starts before the part does; entry nparts = N
struct PartStarts { u64 v[BFQ_MAX_PARTS + 1]; };
__global__ void k_fq_part_index(const FqRec *__restrict__ rec, u64 N, PartStarts ps, int nparts, u64 *__restrict__ idx)
{
    const int p = threadIdx.x;
    if (p > nparts) return;
    u64 lo = 0, hi = N;                                 // first record with hdrStart >= ps.v[p]
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (rec[mid].hdrStart < ps.v[p]) lo = mid + 1; else hi = mid;
    }
    idx[p] = (p == nparts) ? N : lo;
}
__global__ void k_pick_u64(const u64 *__restrict__ src, const u64 *__restrict__ idx, int count, u64 addIdx, u64 *__restrict__ out)
{
    const int p = threadIdx.x;
    if (p < count) out[p] = src[idx[p]] + addIdx * idx[p];
}
void bfq_fastq_part_index(bfq_ctx *c, const DevFastq *fq, const u64 *h_pstart, int nparts, u64 *d_idx)
{
    PartStarts ps;
    for (int p = 0; p <= BFQ_MAX_PARTS; p++) ps.v[p] = p <= nparts ? h_pstart[p] : 0;
    KLAUNCH(c, K_FASTQ, 0.0, k_fq_part_index, 1, 64, (const FqRec *)fq->rec, fq->N, ps, nparts, d_idx);
}
void bfq_pick_u64(bfq_ctx *c, const u64 *d_src, const u64 *d_idx, int count, u64 addIdx, u64 *d_out)
{
    KLAUNCH(c, K_FASTQ, 0.0, k_pick_u64, 1, 64, d_src, d_idx, count, addIdx, d_out);
}
