/*
 * bfqzip_hip.h -- C-ABI of libbfqhip.so: the MI355X (gfx950) implementation of
 * BFQzip's hot path  eBWT build -> positional clusters -> smoothing -> LF inversion.
 *
 * The reference has no in-process API for this path: its boundary is the
 * process boundary between the Python drivers and four executables
 * (BFQzip.py:178-189,206-228; BFQzip_ext.py:165-183,199-220).  The entry points
 * below are what those executables' main() functions reduce to once file I/O is
 * taken out; the drop-in front-ends in bfqzip_amd/csrc/cli/ (gsufsort, eGap,
 * bfq_int, bfq_ext) are thin argv/file wrappers over them.
 *
 * Conventions: plain pointers and sizes, no torch types.  Every function
 * returns 0 on success and a negative BFQ_E_* code on failure;
 * bfq_last_error() gives the message.  A context owns one GPU stream and one
 * device workspace; it is not thread-safe, use one context per thread/GPU.
 * "h_" = host pointer, "d_" = device pointer (hipMalloc'ed / torch storage).
 */
#ifndef BFQZIP_HIP_H
#define BFQZIP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BFQ_OK              0
#define BFQ_E_ARG          -1   /* bad argument                                            */
#define BFQ_E_HIP          -2   /* HIP runtime error / no device                           */
#define BFQ_E_SYMBOL       -3   /* symbol outside {A,C,G,T,N,TERM} (dna_string_n.hpp:87-93) */
#define BFQ_E_NOT_EBWT     -4   /* eBWT does not invert / not in #_i<#_j<A<C<G<N<T order   */
#define BFQ_E_TOO_LONG     -5   /* read longer than BFQ_MAX_READ_LEN                       */
#define BFQ_E_FREQ3        -6   /* three frequent symbols: assert of bfq_int.cpp:505        */
#define BFQ_E_NOMEM        -7
#define BFQ_E_IO           -8   /* file read / write failed (the *_fd entry points)          */

#define BFQ_MAX_READ_LEN 65000  /* LCP is held in 16 bits (reference: LONGEST 10000, bfq_int.cpp:30) */

/* Run-time form of the reference's getopt flags (bfq_int.cpp:883-935) and of its
 * compile-time knobs -DM / -DB (src_int_mem/Makefile:13-23). */
typedef struct bfq_params {
    int32_t K;      /* -k  minimum LCP inside clusters            default 16        */
    int32_t m;      /* -m  minimum cluster length                 default 2         */
    int32_t v;      /* -v  replacement quality (ASCII code, M=2)  default '>' (62)  */
    int32_t f;      /* -f  frequent-symbol percentage             default 40        */
    int32_t t;      /* -t  trusted-quality threshold (phred)      default 20        */
    int32_t term;   /* -s  terminator byte in the eBWT            default '#' (35)  */
    int32_t M;      /* 0 max, 1 mean error, 2 constant, 3 average default 2         */
    int32_t B;      /* 1 = Illumina 8-level binning               default 0         */
    int32_t ext;    /* 1 = bfq_ext arithmetic for M=3 (bfq_ext.cpp:496)             */
    int32_t piles;  /* step 1 pile by pile (first-symbol piles as in bfq_ext.cpp:190-348; 13 n bytes of workspace
                       instead of 28.6 n): 1 always, 0 when the one-piece workspace cannot be had, -1 never;
                       2: the capped mode always (see ws_cap_mib)                                                */
    int32_t ws_cap_mib; /* upper bound of the device workspace in MiB, 0 = none ($BFQ_WS_CAP in bytes, suffixes K/M/G, overrides it).
                       What a block needs: 28.6 n bytes in one piece, 13 n pile by pile; below that the capped mode runs it in
                       ~8 n: two-symbol piles one at a time, edits written to the text position of every row, no eBWT-sized
                       array, no LF table, no inversion -- same bytes out, about a third more time.  Steps 2-4
                       on a GIVEN eBWT (bfq_smooth_invert*: 17 n with the LF table) then run on 64-byte rank blocks answered
                       on demand, qualities edited in place and a replacement array: 7.3 n (LCP given) / 7.8 n + the ring
                       queue of the LCP deduction (deduced: the ring takes what the cap leaves, n/16 entries at least), the
                       eBWT and the qualities themselves included in the count                                            */
    int32_t reserved[5];
} bfq_params;

/* Counters printed by bfq_int.cpp:1004-1019. */
typedef struct bfq_stats {
    uint64_t num_clust, num_clust_discarded, num_clust_amb_discarded, num_clust_mod,
             num_clust_alleq, bases_inside, qs_smoothed, modified;
    uint64_t n_rows, n_reads, n_segments, n_big_segments;
} bfq_stats;

typedef struct bfq_ctx bfq_ctx;

void     bfq_default_params(bfq_params *p);
bfq_ctx *bfq_create(int device, const bfq_params *p);          /* NULL on failure: see bfq_create_error() */
const char *bfq_create_error(void);
void     bfq_destroy(bfq_ctx *c);
int      bfq_set_params(bfq_ctx *c, const bfq_params *p);
const char *bfq_last_error(bfq_ctx *c);
void    *bfq_stream(bfq_ctx *c);                                /* the hipStream_t all kernels run on */
int      bfq_device_count(void);

/* ---- which GPU a one-shot tool uses.  BFQzip_parallel.py:277-285 starts n concurrent `BFQzip.py` children whose
 * gsufsort / bfq_int processes know nothing of each other (BFQzip.py:178-228): the drop-in tools spread over the node's GPUs
 * through per-GPU lease files -- `<dir>/bfqzip_amd.<PCI bus id>.lock`, held by flock() until the process ends; dir =
 * $BFQ_LEASE_DIR, else /dev/shm, else /tmp.  bfq_pick_device() takes the first free GPU (polling when all are busy, so tools
 * on one GPU run one after the other instead of stacking their workspaces) and returns its device index for bfq_create(),
 * or a negative BFQ_E_* code.  $BFQ_DEVICE=k pins the tool to GPU k (still under that GPU's lease); $BFQ_LEASE=0 switches the
 * lease off (device $BFQ_DEVICE or 0); $BFQ_FAKE_DEVICES=n pretends n GPUs (slot k -> device k mod the real count; lease tests
 * on one GPU).  info (may be NULL) receives "device <k> lease <path> waited <s> s".  Long-lived hosts (parallel.py, bench.py)
 * place their ranks themselves and never call it.
 * bfq_device_lease() is the host-only part (no GPU needed): n_slots lock files named by slot_ids[k] (NULL: "slot<k>"),
 * only_slot >= 0 restricts the choice to that slot; returns the slot taken.  bfq_device_release() gives a slot back early. */
int bfq_pick_device(char *info, int info_cap);
int bfq_device_lease(int n_slots, const char *const *slot_ids, int only_slot, char *path_out, int path_cap, double *waited_s);
int bfq_device_release(int slot);

/* ---- phase timeline of a one-shot tool (process + HIP start, lease wait, allocation, file read + H2D, GPU, D2H + file
 * write): bfq_phase(name) closes the running phase and opens `name` (NULL: closes only); bfq_phase_report(tool) prints one
 * line `[bfq phases] {"tool": .., "exec_to_main": s, "<phase>": s, .., "total": s}` on stderr.  Off unless
 * bfq_phase_enable(1) was called (the tools' -V) or $BFQ_TRACE is set; bench.py's dropin_wall_s split is parsed from it. */
void bfq_phase_enable(int on);
void bfq_phase(const char *name);
void bfq_phase_report(const char *tool);

/* ---- step 1: replaces `gsufsort <fq> --bwt --qs -o OUT` (BFQzip.py:184) and
 *      `eGap <fq> --em --mem M --qs -o OUT --lcp --lbytes 1` (BFQzip_ext.py:177).
 * h_bases/h_quals: the reads back to back (lines 2 and 4 of each record),
 * h_read_off[N+1]: offsets.  Outputs (host, n = h_read_off[N]+N entries each):
 * h_bwt, h_bwtqs; h_lcp16 (exact, may be NULL).  term_out: byte written for the
 * terminator ('#' for gsufsort, 0 for eGap). */
int bfq_build_ebwt(bfq_ctx *c, const uint8_t *h_bases, const uint8_t *h_quals,
                   const uint64_t *h_read_off, uint64_t N, int term_out,
                   uint8_t *h_bwt, uint8_t *h_bwtqs, uint16_t *h_lcp16);

/* ---- steps 2-4: replaces `bfq_int -e OUT.bwt -q OUT.bwt.qs -o OUT.fq ...`
 *      (BFQzip.py:215-222; main() bfq_int.cpp:875-1062) when h_lcp is NULL, and
 *      `bfq_ext -e .. -q .. -a OUT.1.lcp ...` (BFQzip_ext.py:208-214) when h_lcp
 *      is given (lcp_bytes = 1, 2 or 4 bytes per entry, little endian).
 * Outputs: h_out_bases/h_out_quals (n - N bytes each), h_out_read_off[N+1].
 * Call bfq_count_reads() first to size them. */
int bfq_count_reads(const uint8_t *h_bwt, uint64_t n, int term, uint64_t *N);
int bfq_smooth_invert(bfq_ctx *c, const uint8_t *h_bwt, const uint8_t *h_bwtqs,
                      const void *h_lcp, int lcp_bytes, uint64_t n,
                      uint8_t *h_out_bases, uint8_t *h_out_quals, uint64_t *h_out_read_off,
                      bfq_stats *st);

/* ---- fused path (no intermediate files): reads in -> smoothed reads out. */
int bfq_run_reads(bfq_ctx *c, const uint8_t *h_bases, const uint8_t *h_quals,
                  const uint64_t *h_read_off, uint64_t N,
                  uint8_t *h_out_bases, uint8_t *h_out_quals, bfq_stats *st);

/* Same with everything resident in HBM (what bench.py times).  d_out_* may
 * alias nothing else; total = number of bases = d_read_off[N] (given by the
 * caller so that no synchronising read-back is needed). */
int bfq_run_reads_device(bfq_ctx *c, const uint8_t *d_bases, const uint8_t *d_quals,
                         const uint64_t *d_read_off, uint64_t N, uint64_t total,
                         uint8_t *d_out_bases, uint8_t *d_out_quals, bfq_stats *st);

/* ---- FASTQ text in / out, parsed and formatted on the GPU (host buffers).
 * Records are 4 lines; CR before LF is dropped from lines 2 and 4; a record whose quality line is
 * not as long as its sequence is an error (checkFASTQ.py:18-32).
 *   bfq_fastq_build_ebwt   : gsufsort / eGap on the file's bytes; outputs hold cap_rows entries
 *                            (len/2 + 1 is always enough); *n_rows / *n_reads receive the sizes.
 *   bfq_fastq_run          : the whole path, text to text; keep_headers != 0 passes every record's
 *                            header line through (BFQzip.py --headers), else "@" (bfq_int.cpp:758,805).
 *   bfq_smooth_invert_fastq: bfq_int / bfq_ext writing the FASTQ text itself; h_headers = the -H file
 *                            (one line per read) or NULL.
 *   bfq_fastq_run_streams  : the whole path with the result as the separate streams that BFQzip.py's
 *                            --m2/--m3 modes compress (BFQzip.py:19-21,192-251): h_dna = `sed -n 2~4p OUT.fq`,
 *                            h_qs = `sed -n 4~4p OUT.fq` (total bases + reads bytes each), h_hdr = `sed -n 1~4p
 *                            in.fastq` (may be NULL).  A capacity of `len` is always enough for each.
 * Output size: bfq_fastq_out_bound(total bases, reads, header bytes without newlines or 0). */
uint64_t bfq_fastq_out_bound(uint64_t total_bases, uint64_t n_reads, uint64_t header_bytes);
int bfq_fastq_build_ebwt(bfq_ctx *c, const uint8_t *h_fastq, uint64_t len, int term_out,
                         uint8_t *h_bwt, uint8_t *h_bwtqs, uint16_t *h_lcp16, uint64_t cap_rows,
                         uint64_t *n_rows, uint64_t *n_reads);
int bfq_fastq_run(bfq_ctx *c, const uint8_t *h_fastq, uint64_t len, int keep_headers,
                  uint8_t *h_out, uint64_t cap, uint64_t *out_len, bfq_stats *st);
int bfq_fastq_run_streams(bfq_ctx *c, const uint8_t *h_fastq, uint64_t len,
                          uint8_t *h_dna, uint8_t *h_qs, uint64_t cap_stream, uint64_t *stream_len,
                          uint8_t *h_hdr, uint64_t cap_hdr, uint64_t *hdr_len, bfq_stats *st);
int bfq_smooth_invert_fastq(bfq_ctx *c, const uint8_t *h_bwt, const uint8_t *h_bwtqs,
                            const void *h_lcp, int lcp_bytes, uint64_t n,
                            const uint8_t *h_headers, uint64_t headers_len,
                            uint8_t *h_out, uint64_t cap, uint64_t *out_len, bfq_stats *st);

/* ---- the two tools on open files (what the drop-in executables call): same operations as bfq_fastq_build_ebwt
 * and bfq_smooth_invert_fastq, the bytes moved between the files and the GPU by the library's pinned staging
 * pipeline (pread / pwrite by several threads: no mapping, no intermediate copy of a whole file in host memory).
 * Outputs are written from offset 0 of descriptors the caller opened for writing (and truncated); a descriptor
 * < 0 = not wanted / not given.  lcp_bytes: 1, 2 or 4 (eGap --lbytes; 1 saturates at 255). */
int bfq_fastq_build_ebwt_fd(bfq_ctx *c, int fastq_fd, uint64_t len, int term_out, int bwt_fd, int bwtqs_fd,
                            int lcp_fd, int lcp_bytes, uint64_t *n_rows, uint64_t *n_reads);
int bfq_smooth_invert_fastq_fd(bfq_ctx *c, int bwt_fd, int bwtqs_fd, int lcp_fd, int lcp_bytes, uint64_t n,
                               int headers_fd, uint64_t headers_len, int out_fd, uint64_t *out_len, bfq_stats *st);
/* What a tool can start before the GPU is initialised: bfq_output_prefault() sizes an output descriptor to map_len bytes,
 * maps it and lets helper threads fault in its first prefault_len bytes in the background (the page-cache pages of a 9 GB
 * output are allocated and zeroed by the kernel at ~6 GB/s; done beside the upload and the GPU work, the final copy runs at
 * memcpy speed).  The *_fd entry point that is later handed the same descriptor picks the mapping up and cuts the file to
 * its real length.  Bounds: eBWT / QS files <= len / 2 + 64 bytes for a FASTQ of len bytes, bfq_fastq_rows_estimate() =
 * the likely row count (from the first records of the file); the FASTQ text bfq_int writes is >= 2 n and <= 6 n + the
 * header file + 4096 bytes for an eBWT of n rows.  Optional: without it the entry points do the same from their first line. */
int      bfq_output_prefault(int fd, uint64_t map_len, uint64_t prefault_len);
uint64_t bfq_fastq_rows_estimate(int fastq_fd, uint64_t len);

/* ---- one block of BFQzip_parallel.py as one call (BFQzip_parallel.py:277-285 runs `BFQzip.py <block> --rebuild -0
 * [--headers]` per block; :325-360 appends mate block k of file 2 to block k of file 1; :153-172 cuts the
 * block's output back into OUT_1 / OUT_2 by line count).
 * The block's text is given as 1..BFQ_MAX_PARTS byte ranges (mmap'ed file ranges or pinned buffers; a part that
 * lacks its final newline gets one) processed as ONE collection, parts in order.  Any of the outputs may be
 * asked for in the same pass: the FASTQ text (out_fastq), the --m2/--m3 streams (out_dna, out_qs, out_hdr).
 * part_*[p] = where part p's share of each output starts (entry nparts = the end), part_reads[p] = index of its
 * first read.  Pinned host buffers (bfq_host_alloc) are transferred by direct DMA, pageable ones through the
 * library's pinned staging pipeline. */
#define BFQ_MAX_PARTS 4
typedef struct bfq_text_part { const uint8_t *data; uint64_t len; } bfq_text_part;
typedef struct bfq_fastq_job {
    const bfq_text_part *parts; int32_t nparts;
    int32_t keep_headers;                     /* FASTQ text: header lines verbatim (BFQzip.py --headers) or "@"  */
    uint8_t *out_fastq; uint64_t cap_fastq;   /* NULL: not wanted; bfq_fastq_out_bound() / input length + 16     */
    uint8_t *out_dna, *out_qs; uint64_t cap_stream;   /* total bases + reads bytes each                           */
    uint8_t *out_hdr; uint64_t cap_hdr;       /* `sed -n 1~4p` of the input                                      */
    /* results */
    uint64_t fastq_len, stream_len, hdr_len, n_reads, total_bases;
    uint64_t part_reads[BFQ_MAX_PARTS + 1], part_fastq_off[BFQ_MAX_PARTS + 1],
             part_stream_off[BFQ_MAX_PARTS + 1], part_hdr_off[BFQ_MAX_PARTS + 1];
    /* steps 1-5 in one call: the streams leave as BFQRANS2 containers (bfq_stream_compress, below) instead of raw bytes --
     * what `BFQzip.py --m2/--m3` without -0 produces through 7z / bsc (BFQzip.py:253-275).  The raw streams never cross
     * the bus.  stream_len / hdr_len stay the RAW lengths; *_bytes = what was written to out_dna / out_qs / out_hdr.
     * Capacities: bfq_stream_bound(raw length) always suffices (a tiny stream's container is larger than the stream; the raw
     * length is enough from a few MB on); out_dna of modes 2 / 3 holds two containers: twice that + 40. */
    int32_t  compress_streams;                /* 1: as described; 2: eBWT-domain containers (bfq_stream_ebwt_decode, below);
                                                 3: the same with the qualities in read order (smallest output) */
    int32_t  reserved0;
    uint64_t dna_bytes, qs_bytes, hdr_bytes;
} bfq_fastq_job;
int bfq_fastq_run_job(bfq_ctx *c, bfq_fastq_job *job, bfq_stats *st);

/* ---- one collection over several GPUs with the UNSHARDED result (opt-in; not in the reference, whose parallel driver
 * gives up the clusters that span blocks, README.md:107).  The per-GPU pieces; bfqzip_amd/parallel.py --global holds the
 * device buffers (torch tensors) and runs the collectives.  d_ = device pointers.
 *   bfq_glob_begin      : upload and parse this rank's block; its text stays resident in the context
 *   bfq_glob_local_text : the block as terminated text: symbol codes (# 0, A 1, C 2, G 3, N 4, T 5) and qualities,
 *                         total_bases + n_reads bytes each -> exchanged so that every rank holds the whole text (n bytes)
 *   bfq_glob_pile_counts: counts36[6 s + s2] = suffixes that start with symbols s, s2 (row 0: the terminator suffixes)
 *   bfq_glob_init_out   : line-stream copy of the whole text (letters / qualities, '\n' at the terminators)
 *   bfq_glob_run_pile   : sort + refine the pile (s, s2) of the global eBWT, cluster analysis, edits written to d_sym / d_qual
 *                         at the text position each row stands for; statistics of this pile in *st
 *   bfq_glob_finish     : this block's line streams (after the exchange) -> FASTQ text / streams as in bfq_fastq_run_job
 *                         (job->parts ignored; qualities are binned here when B = 1) */
int bfq_glob_begin(bfq_ctx *c, const bfq_text_part *parts, int nparts, uint64_t *n_reads, uint64_t *total_bases);
int bfq_glob_local_text(bfq_ctx *c, uint8_t *d_T8, uint8_t *d_Q8);
int bfq_glob_pile_counts(bfq_ctx *c, const uint8_t *d_T8, uint64_t n, uint64_t *counts36);
int bfq_glob_init_out(bfq_ctx *c, const uint8_t *d_T8, const uint8_t *d_Q8, uint64_t n, uint8_t *d_sym, uint8_t *d_qual);
int bfq_glob_run_pile(bfq_ctx *c, const uint8_t *d_T8, const uint8_t *d_Q8, uint64_t n, int s, int s2,
                      uint8_t *d_sym, uint8_t *d_qual, bfq_stats *st);
int bfq_glob_finish(bfq_ctx *c, uint8_t *d_dna, uint8_t *d_qs, bfq_fastq_job *job);

/* Pinned (page-locked) host memory for the buffers above. */
void *bfq_host_alloc(uint64_t bytes);
void  bfq_host_free(void *p);

/* Host-side line index of a text (what BFQzip_parallel.py:295-319 does with Python line loops): counts[i] =
 * number of '\n' in bytes [i*chunk, (i+1)*chunk), computed by `threads` threads (0: default);
 * bfq_text_nth_newline = offset of the k-th (0-based) '\n' of the range or -1.  Pure host functions (no GPU). */
int     bfq_text_count_lines(const uint8_t *h_text, uint64_t len, uint64_t chunk, uint64_t *counts, int threads);
/* a host buffer into a file at an offset by several threads (fallocate + shared mapping; pwrite when the file cannot be
 * mapped): how the multi-GPU driver puts every block's outputs at their final place of the shared output files
 * (BFQzip_parallel.py:137-179 merges with `cat`).  fd must be open for reading and writing; the file grows as needed and is
 * never shrunk, so several processes may fill different ranges of one file.  threads 0: by the CPU budget.  Host only. */
int     bfq_file_put(int fd, uint64_t offset, const void *src, uint64_t len, int threads);
/* ... or the range itself as memory to fill (allocated, mapped, populated by a few threads): an output buffer for
 * bfq_fastq_run_job / bfq_glob_finish that IS the file -- no copy afterwards.  NULL when the file cannot be mapped;
 * bfq_file_unmap() takes the same offset and length. */
void   *bfq_file_map(int fd, uint64_t offset, uint64_t len, int threads);
int     bfq_file_unmap(void *p, uint64_t offset, uint64_t len);
int64_t bfq_text_nth_newline(const uint8_t *h_text, uint64_t len, uint64_t k);

/* Device-resident eBWT of the last bfq_run_reads*() / bfq_build_ebwt() call
 * (valid until the next call on the context): copies to host. Any may be NULL.
 * h_bwtqs receives the permuted qualities as built (before smoothing). */
int bfq_fetch_ebwt(bfq_ctx *c, uint8_t *h_bwt, uint8_t *h_bwtqs, uint16_t *h_lcp16);

/* ---- synthetic reads (seeded, counter based; DESIGN.md "synthetic workload").
 * Fixed length L when Lmin == Lmax.  Host and device versions produce identical
 * bytes.  read_off[N+1] is written too. */
typedef struct bfq_synth {
    uint64_t seed;
    uint64_t N;          /* reads                                   */
    uint32_t Lmin, Lmax; /* read length range (inclusive)           */
    uint32_t coverage;   /* genome length = N*Lavg/coverage         */
    uint32_t err_ppm;    /* substitution errors, per million bases  */
    uint32_t n_ppm;      /* 'N' calls, per million bases            */
    uint32_t snp_every;  /* haplotype SNP period (~1000)            */
    uint32_t dsnp_every; /* adjacent double-SNP period (~10000)     */
    uint32_t both_strands;
    uint64_t first;      /* these N reads are reads [first, first+N) ...                          */
    uint64_t collection; /* ... of a collection of this many reads (0: N; sets the genome length): */
    uint32_t reserved[2];/*     a block of BFQzip_parallel's split can be generated on its own     */
} bfq_synth;
void bfq_synth_default(bfq_synth *s, uint64_t N, uint32_t L);
uint64_t bfq_synth_total(const bfq_synth *s);                    /* total bases = read_off[N] */
int bfq_synth_host(const bfq_synth *s, uint8_t *h_bases, uint8_t *h_quals, uint64_t *h_read_off);
int bfq_synth_device(bfq_ctx *c, const bfq_synth *s, uint8_t *d_bases, uint8_t *d_quals,
                     uint64_t *d_read_off);
/* The same reads as the text of a FASTQ file (header lines "@SYN.<read number>", "+" lines bare), generated and
 * formatted on the device, copied to h_out (cap >= N * (2 * Lmax + 30) is always enough). */
int bfq_synth_fastq(bfq_ctx *c, const bfq_synth *s, uint8_t *h_out, uint64_t cap, uint64_t *out_len);

/* ---- stream codec: entropy coding of OUT.fq.dna / OUT.fq.qs / OUT.h on the GPU.
 * Replaces step 5 of the reference, which hands every stream to an external tool: `7z a -mm=PPMd <f>.7z <f>`
 * (step5, BFQzip.py:253-263) or `external/libbsc/bsc e <f> <f>.bsc -T` (step5b, BFQzip.py:265-275).  The front-end
 * dropin/external/libbsc/bsc takes that command line (`bsc e IN OUT [options]`, `bsc d IN OUT`).
 * The containers are this project's own -- neither 7z nor libbsc are part of the reference tree -- and are stated in
 * oracle/bfq_codec_ref.c: "BFQRANS2" (any bytes: static order-k model + range-ANS, segments of 8192 symbols), "BFQLINE1"
 * (read names: a line-delta transform in front of it) and "BFQDNAC1" (read-order DNA, lines of A C G T N: a hashed
 * order-K context model that adapts block by block, rebuilt by the decoder from what it has decoded; a third of the static
 * container's size at 30x coverage, a quarter of its speed; $BFQ_DNA_STATIC=1 keeps the static one).
 * Any bytes compress; host buffers in and out.  bfq_stream_decompress takes containers of any kind back to back. */
uint64_t bfq_stream_bound(uint64_t len);                          /* capacity that always suffices for `len` raw bytes */
int64_t  bfq_stream_raw_len(const uint8_t *h_in, uint64_t len);   /* raw length of a container, -1 if it is not one  */
int bfq_stream_compress(bfq_ctx *c, const uint8_t *h_in, uint64_t len, uint8_t *h_out, uint64_t cap, uint64_t *out_len);
int bfq_stream_decompress(bfq_ctx *c, const uint8_t *h_in, uint64_t len, uint8_t *h_out, uint64_t cap, uint64_t *out_len);
/* eBWT-domain containers (bfq_fastq_job.compress_streams = 2; out_fastq must be NULL): out_dna receives "BFQEBWT1" |
 * u64 rows | u64 reads | u32 terminator byte | u32 flags | u64 bytes of the next container | the container of the eBWT's
 * symbols AFTER noise reduction | the container of the replaced rows' original symbols (0 elsewhere), out_qs the
 * container of the rows' qualities after smoothing (row order, n = bases + reads bytes each).  In row order the symbols of a
 * deep collection are runs -- half the size of the read-order stream -- and the compressing side skips the inversion.
 * compress_streams = 3 (flags bit 0) keeps the qualities in READ order (OUT.fq.qs as in mode 1: they code better along the
 * read) at the price of one walk on the compressing side.
 * bfq_stream_ebwt_decode inverts them back to the line streams OUT.fq.dna / OUT.fq.qs (cap >= rows bytes each). */
int bfq_stream_ebwt_decode(bfq_ctx *c, const uint8_t *h_bwtz, uint64_t len_b, const uint8_t *h_qsz, uint64_t len_q,
                           uint8_t *h_dna, uint8_t *h_qs, uint64_t cap, uint64_t *stream_len, uint64_t *n_reads);
/* device-resident form (input and output in device memory): bfq_stream_reserve(len) sizes the workspace once */
int bfq_stream_reserve(bfq_ctx *c, uint64_t len);
int bfq_stream_compress_device(bfq_ctx *c, const uint8_t *d_in, uint64_t len, uint8_t *d_out, uint64_t cap, uint64_t *out_len);

/* ---- profiling: per-kernel HIP-event times accumulated over the calls since
 * the last bfq_prof_reset() (events recorded on bfq_stream()). */
int  bfq_prof_enable(bfq_ctx *c, int on);
void bfq_prof_reset(bfq_ctx *c);
int  bfq_prof_count(bfq_ctx *c);
int  bfq_prof_get(bfq_ctx *c, int idx, char *name, int name_cap, double *total_ms,
                  uint64_t *launches, double *alg_bytes_total);
/* per-launch durations (ms, launch order, since the last reset) of ONE kernel chosen by its bfq_prof_get index (-1: none):
 * bfq_prof_trace copies up to cap of them and returns how many there are -- e.g. the radix passes one by one */
int     bfq_prof_trace_select(bfq_ctx *c, int idx);
int64_t bfq_prof_trace(bfq_ctx *c, float *ms, uint64_t cap);

uint64_t bfq_workspace_bytes(bfq_ctx *c);      /* current device workspace size */
const char *bfq_version(void);

#ifdef __cplusplus
}
#endif
#endif
