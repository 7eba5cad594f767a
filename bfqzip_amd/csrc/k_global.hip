// k_global.hip -- ONE collection over several GPUs with the unsharded result (opt-in).
//
// BFQzip_parallel.py trades compression for speed: every block gets its own eBWT, so clusters never see the reads of
// other blocks (README.md:107 of the reference).  The global mode keeps the speed-up and gives the result of the
// UNSHARDED run (`BFQzip.py in.fastq`, one eBWT over everything), bit for bit:
//   1. every rank parses its block (the same split) and contributes its part of the terminated text (symbol codes +
//      qualities, 2 bytes per base); the parts are exchanged so that every GPU holds the whole text (RCCL broadcast /
//      all-gather: 2 n bytes, 9 GB at 30 M x 150);
//   2. the suffixes are partitioned into piles by their first two symbols (k_piles.hip) and the piles are dealt to the
//      ranks by size; a rank sorts and refines ITS piles of the global eBWT.  Clusters never cross a pile boundary
//      (the LCP there is at most 1 < K), so the cluster analysis of a pile is local -- with one exception the reference
//      resolves through bwt[LF(row)] and this mode through the text: the symbol before the preceding one;
//   3. every row knows the text position of its suffix (the sort payload), so k_cluster writes its edits straight to
//      that position of a line-stream copy of the whole text ("position mode", k_cluster.hip);
//   4. the ranks' copies differ from the original only where their own piles edited: the differences are combined with
//      one all-reduce (sum of byte-wise XOR deltas, disjoint support) and every rank formats its own block.
// The entry points below are the per-GPU pieces; bfqzip_amd/parallel.py (--global) holds the buffers as torch tensors
// and runs the collectives.  All pointers named d_ are device pointers.
#include <string.h>
#include "bfq_internal.h"
#include "bfq_device.h"

u64 bfq_fastq_count_lines(bfq_ctx *c, const u8 *d_buf, u64 len);   // k_fastq.hip

template <class F> static int guarded_g(bfq_ctx *c, F body)
{
    if (!c) return BFQ_E_ARG;
    try {
        HIP_CHECK(hipSetDevice(c->device));
        c->err.clear();
        body();
        return BFQ_OK;
    } catch (const BfqError &e) {
        c->err = e.msg;
        (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
        c->recs.clear(); c->evUsed = 0;
        return e.code;
    } catch (const std::bad_alloc &) {
        c->err = "host out of memory";
        return BFQ_E_NOMEM;
    }
}

// the resident block text parsed again (cheap: a few streaming kernels); arena sized for the parse only
static void reparse(bfq_ctx *c, DevFastq *fq, size_t extra)
{
    if (!c->residentValid) throw BfqError{BFQ_E_ARG, "no block text resident: call bfq_glob_begin first"};
    const u64 len = c->residentLen;
    c->reserve(16 * (len / 4096 + 16) + (64u << 20));
    const u64 nlines = bfq_fastq_count_lines(c, c->d_text, len);
    c->reserve(3 * (len + 4096) + 128 * (nlines / 4 + 64) + 8 * (nlines + 64) + extra + (64u << 20));
    c->zeroCounters();
    bfq_fastq_parse(c, c->d_text, len, fq);
}

void bfq_fastq_part_index(bfq_ctx *c, const DevFastq *fq, const u64 *h_pstart, int nparts, u64 *d_idx);   // k_fastq.hip
void bfq_pick_u64(bfq_ctx *c, const u64 *d_src, const u64 *d_idx, int count, u64 addIdx, u64 *d_out);      // k_fastq.hip

extern "C" int bfq_glob_begin(bfq_ctx *c, const bfq_text_part *parts, int nparts, uint64_t *n_reads, uint64_t *total_bases)
{
    return guarded_g(c, [&] {
        if (nparts < 1 || nparts > BFQ_MAX_PARTS || !parts) throw BfqError{BFQ_E_ARG, "1..BFQ_MAX_PARTS parts"};
        u64 len = 0;
        std::vector<u8> addNl(nparts, 0);
        for (int p = 0; p < nparts; p++) {
            if (parts[p].len && !parts[p].data) throw BfqError{BFQ_E_ARG, "null FASTQ text"};
            len += parts[p].len;
            if (parts[p].len && parts[p].data[parts[p].len - 1] != (u8)'\n') { addNl[p] = 1; len++; }
        }
        u8 *d = c->textBuf(len + 64);
        u64 o = 0;
        for (int p = 0; p < nparts; p++) {
            c->residentPstart[p] = o;
            bfq_upload(c, d + o, parts[p].data, parts[p].len);
            o += parts[p].len;
            if (addNl[p]) { HIP_CHECK(hipMemsetAsync(d + o, '\n', 1, c->stream)); o++; }
        }
        c->residentPstart[nparts] = len;
        c->residentLen = len; c->residentValid = true; c->residentParts = nparts;
        DevFastq fq;
        reparse(c, &fq, 0);
        // reads and bases of every part (n_reads / total_bases: nparts entries each)
        u64 *d_pidx = c->alloc<u64>(nparts + 1), *d_pb = c->alloc<u64>(nparts + 1);
        bfq_fastq_part_index(c, &fq, c->residentPstart, nparts, d_pidx);
        bfq_pick_u64(c, fq.roff, d_pidx, nparts + 1, 0, d_pb);
        u64 hidx[BFQ_MAX_PARTS + 1], hb[BFQ_MAX_PARTS + 1];
        HIP_CHECK(hipMemcpyAsync(hidx, d_pidx, 8 * (nparts + 1), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipMemcpyAsync(hb, d_pb, 8 * (nparts + 1), hipMemcpyDeviceToHost, c->stream));
        c->fetchCounters();
        for (int p = 0; p < nparts; p++) {
            if (n_reads) n_reads[p] = hidx[p + 1] - hidx[p];
            if (total_bases) total_bases[p] = hb[p + 1] - hb[p];
        }
    });
}

extern "C" int bfq_glob_local_text(bfq_ctx *c, uint8_t *d_T8, uint8_t *d_Q8)
{
    return guarded_g(c, [&] {
        DevFastq fq;
        reparse(c, &fq, 0);
        const u64 n = fq.total + fq.N;
        if (!n) return;
        size_t mk = c->mark();
        u64 nwords = n / BFQ_SYMS_PER_WORD + 3;
        u64 *text3 = c->alloc<u64>(bfq_t3_alloc(nwords));                     // by-product of the existing text builder, not used here
        u8 *T8 = c->alloc<u8>(n + 64), *Q8 = c->alloc<u8>(n + 64);
        bfq_build_text(c, fq.bases, fq.quals, fq.roff, fq.N, n, T8, Q8, text3, nwords);
        HIP_CHECK(hipMemcpyAsync(d_T8, T8, n, hipMemcpyDeviceToDevice, c->stream));
        HIP_CHECK(hipMemcpyAsync(d_Q8, Q8, n, hipMemcpyDeviceToDevice, c->stream));
        c->fetchCounters();
        c->release(mk);
        if (c->h_cnt.errSymbol) throw BfqError{BFQ_E_SYMBOL, "symbol outside {A,C,G,T,N,terminator}"};
        if (c->h_cnt.errTooLong) throw BfqError{BFQ_E_TOO_LONG, "read longer than BFQ_MAX_READ_LEN"};
    });
}

extern "C" int bfq_glob_pile_counts(bfq_ctx *c, const uint8_t *d_T8, uint64_t n, uint64_t *counts36)
{
    return guarded_g(c, [&] {
        if (!counts36) throw BfqError{BFQ_E_ARG, "null counts"};
        c->reserve(40 * (n / BFQ_RS_BLOCK_ELEMS + 64) * 8 + (64u << 20));
        memset(counts36, 0, 36 * sizeof(uint64_t));
        if (n) bfq_pile_pair_counts(c, d_T8, n, (u64 *)counts36);
        c->globN = n;                                           // bfq_glob_run_pile sizes its workspace from these
        memcpy(c->globCounts, counts36, sizeof c->globCounts);
    });
}

// line-stream form of a terminated text: letters and '\n' instead of codes, '\n' instead of the terminators' quality slots
__global__ __launch_bounds__(256) void k_glob_init_out(const u8 *__restrict__ T8, const u8 *__restrict__ Q8, u64 n, u8 *__restrict__ sym, u8 *__restrict__ qual)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 code = T8[i] & 7u;
        sym[i] = code ? bfq_code_sym(code) : (u8)10;
        qual[i] = code ? Q8[i] : (u8)10;
    }
}
extern "C" int bfq_glob_init_out(bfq_ctx *c, const uint8_t *d_T8, const uint8_t *d_Q8, uint64_t n, uint8_t *d_sym, uint8_t *d_qual)
{
    return guarded_g(c, [&] {
        if (n) KLAUNCH(c, K_MISC, 4.0 * (double)n, k_glob_init_out, bfq_grid(n, 256), 256, d_T8, d_Q8, n, d_sym, d_qual);
        c->sync();
        c->profCollect();
    });
}

extern "C" int bfq_glob_run_pile(bfq_ctx *c, const uint8_t *d_T8, const uint8_t *d_Q8, uint64_t n, int s, int s2, uint8_t *d_sym,
                                 uint8_t *d_qual, bfq_stats *st)
{
    return guarded_g(c, [&] {
        if (st) memset(st, 0, sizeof *st);
        if (s < 1 || s > 5 || s2 < 0 || s2 > 5) throw BfqError{BFQ_E_ARG, "pile symbols: first 1..5 (A C G N T), second 0..5"};
        if (c->P.K < 2) throw BfqError{BFQ_E_ARG, "global mode needs -k >= 2 (clusters must not cross the two-symbol piles)"};
        if (n >= (1ull << BFQ_POS_BITS)) throw BfqError{BFQ_E_ARG, "collection too large (2^37 rows)"};
        // text3 + block counts + the records of this pile: its size is known when bfq_glob_pile_counts ran on this text (low-
        // complexity or amplicon libraries put far more than the usual n / 16 into one pile); else sized generously from n
        const u64 known = (c->globN == n) ? c->globCounts[6 * s + s2] : 0;
        const u64 cap = (known ? known : n / 4) + (1u << 20);
        c->reserve(8 * bfq_t3_alloc(n / 21 + 3) + 40 * (n / BFQ_RS_BLOCK_ELEMS + 64) * 8 + 30 * (cap + 256) + 12 * 256 * (ceil_div(cap + 1, bfq_radix_block_elems(cap)) + 8200) + (cap + 4096) / 32768 * 64 + (128u << 20));
        c->zeroCounters();
        const u64 nwords = n / BFQ_SYMS_PER_WORD + 3;
        u64 *text3 = c->alloc<u64>(bfq_t3_alloc(nwords));
        bfq_pack_text(c, d_T8, n, text3, nwords);
        PileRows pr;
        c->n = 0; c->N = 0;
        const u64 m = bfq_run_one_pile(c, d_T8, d_Q8, text3, n, (u32)s, (u32)s2, c->P.term & 0xFF, &pr);
        if (m) {
            u8 *in = c->alloc<u8>(m + 64);
            bfq_lcp_flags(c, pr.lcp, m, c->P.K, in);
            ClusterPos pm{pr.w12, text3, d_sym, d_qual, 0};      // binning is applied when the block is written (bfq_glob_finish)
            RankIndex none{nullptr, m};
            bfq_clusters(c, none, pr.bwt, pr.qs, in, m, &pm);
        }
        c->fetchCounters();
        c->profCollect();
        const DevCounters &h = c->h_cnt;
        if (h.errFreq3) throw BfqError{BFQ_E_FREQ3, "three frequent symbols in a cluster (bfq_int.cpp:505 assert); raise -f"};
        if (st) {
            const u64 *x = h.stats;
            st->num_clust = x[0]; st->num_clust_discarded = x[1]; st->num_clust_amb_discarded = x[2]; st->num_clust_mod = x[3];
            st->num_clust_alleq = x[4]; st->bases_inside = x[5]; st->qs_smoothed = x[6]; st->modified = x[7];
            st->n_rows = m; st->n_segments = h.nSegs; st->n_big_segments = h.bigTotal + h.bigCount;
        }
    });
}

__global__ __launch_bounds__(256) void k_bin_lines(u8 *__restrict__ qs, u64 n)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 q = qs[i];
        if (q != 10u) qs[i] = (u8)bfq_bin8(q);
    }
}

// d_dna / d_qs: this block's line streams after the exchange (qualities not binned yet; d_qs is binned in place when B = 1).
// Outputs as in bfq_fastq_run_job (job->parts / nparts are ignored: the resident block text supplies headers, lengths and
// the parts whose shares of every output are reported).
extern "C" int bfq_glob_finish(bfq_ctx *c, uint8_t *d_dna, uint8_t *d_qs, bfq_fastq_job *J)
{
    return guarded_g(c, [&] {
        if (!J) throw BfqError{BFQ_E_ARG, "null job"};
        DevFastq fq;
        reparse(c, &fq, 0);
        const u64 sl = fq.total + fq.N;
        J->n_reads = fq.N; J->total_bases = fq.total;
        J->fastq_len = J->stream_len = J->hdr_len = 0;
        J->dna_bytes = J->qs_bytes = J->hdr_bytes = 0;               // (compress_streams is not offered here: raw streams)
        if (c->P.B && sl) KLAUNCH(c, K_MISC, 2.0 * (double)sl, k_bin_lines, bfq_grid(sl, 256), 256, d_qs, sl);
        const int np = c->residentParts;
        u64 *d_pidx = c->alloc<u64>(np + 1), *d_pick = c->alloc<u64>(4 * (np + 1));
        bfq_fastq_part_index(c, &fq, c->residentPstart, np, d_pidx);
        std::vector<u64> hp(4 * (np + 1), 0);
        bool pickF = false, pickH = false;
        HIP_CHECK(hipMemcpyAsync(hp.data(), d_pidx, 8 * (np + 1), hipMemcpyDeviceToHost, c->stream));
        bfq_pick_u64(c, fq.roff, d_pidx, np + 1, 1, d_pick + 2 * (np + 1));               // roff[i] + i
        if (J->out_hdr) {
            u8 *d_hdr = nullptr;
            u64 *hOff = nullptr;
            u64 hl = 0;
            bfq_fastq_hdr_stream(c, fq.N, c->d_text, &fq, &d_hdr, &hl, &hOff);
            J->hdr_len = J->hdr_bytes = hl;
            if (hl > J->cap_hdr) throw BfqError{BFQ_E_ARG, "stream buffer too small"};
            bfq_download(c, J->out_hdr, d_hdr, hl);
            bfq_pick_u64(c, hOff, d_pidx, np + 1, 0, d_pick + 3 * (np + 1));
            pickH = true;
        }
        if (J->out_dna || J->out_qs) {
            if (sl > J->cap_stream) throw BfqError{BFQ_E_ARG, "stream buffer too small"};
            J->stream_len = sl;
            if (J->out_dna) { bfq_download(c, J->out_dna, d_dna, sl); J->dna_bytes = sl; }
            if (J->out_qs) { bfq_download(c, J->out_qs, d_qs, sl); J->qs_bytes = sl; }
        }
        if (J->out_fastq) {
            u8 *d_out = nullptr;
            u64 *recOff = nullptr;
            u64 ol = bfq_fastq_format(c, d_dna, d_qs, fq.roff, fq.N, J->keep_headers ? 2 : 0, c->d_text, c->residentLen, &fq, &d_out, &recOff, true);
            J->fastq_len = ol;
            if (ol > J->cap_fastq) throw BfqError{BFQ_E_ARG, "output buffer smaller than the FASTQ text"};
            bfq_download(c, J->out_fastq, d_out, ol);
            bfq_pick_u64(c, recOff, d_pidx, np + 1, 0, d_pick + (np + 1));
            pickF = true;
        }
        HIP_CHECK(hipMemcpyAsync(hp.data() + (np + 1), d_pick + (np + 1), 8 * 3 * (np + 1), hipMemcpyDeviceToHost, c->stream));
        c->fetchCounters();
        c->profCollect();
        for (int p = 0; p <= BFQ_MAX_PARTS; p++) J->part_reads[p] = J->part_fastq_off[p] = J->part_stream_off[p] = J->part_hdr_off[p] = 0;
        for (int p = 0; p <= np; p++) {
            J->part_reads[p] = hp[p];
            J->part_fastq_off[p] = pickF ? hp[(np + 1) + p] : 0;
            J->part_stream_off[p] = hp[2 * (np + 1) + p];
            J->part_hdr_off[p] = pickH ? hp[3 * (np + 1) + p] : 0;
        }
    });
}
