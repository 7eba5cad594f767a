// bfq_internal.h -- context, workspace arena, profiling and stage entry points
// shared by the .hip translation units of libbfqhip.so.  Not part of the ABI.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>
#include <thread>
#include "../../include/bfqzip_hip.h"
#include "bfq_common.h"
#include "bfq_internal_host.h"

struct BfqError {
    int code;
    std::string msg;
};

#define HIP_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            throw BfqError{BFQ_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)};     \
    } while (0)

// ---- kernel ids for the profiler ------------------------------------------------
enum BfqKernel {
    K_TEXT = 0, K_PACK, K_KEYS, K_RADIX_HIST, K_SCAN, K_RADIX_SCATTER, K_HUGE_ROUND, K_CLUSTER_BIG,
    K_REFINE_WAVE, K_REFINE_BIG, K_EMIT, K_RANK_BUILD, K_RANK_FINAL, K_LCP_FLAGS, K_CLUSTER,
    K_INVERT_COUNT, K_INVERT, K_SYNTH, K_FASTQ, K_BFS, K_CODEC, K_MISC, K_NUM
};
extern const char *const BFQ_KERNEL_NAMES[K_NUM];

struct ProfRec { int id; hipEvent_t a, b; double bytes; };

#define BFQ_IO_MAX_WORKERS 16
#define BFQ_IO_STAGE_BYTES (16u << 20)

// device-side counters / small outputs read back by the host (one hipMemcpy)
struct DevCounters {
    u64 stats[8];        // bfq_stats cluster counters, same order
    u64 bigCount;        // number of segments > 64 rows
    u64 errSymbol;       // >0: forbidden symbol met
    u64 errInvert;       // >0: LF walk did not close
    u64 errFreq3;        // >0: three frequent symbols in a cluster
    u64 errTooLong;      // >0: read longer than BFQ_MAX_READ_LEN
    u64 tot[6];          // symbol totals of the eBWT: # A C G N T
    u64 mismatch;        // >0: rebuilt eBWT differs from the given one
    u64 nSegs;           // segments of >= 2 rows met by the refinement
    u64 errFastq;        // >0: a record whose quality line is not as long as its sequence
    u64 hugeCount;       // segments above BFQ_HUGE_SEG rows, left to the radix rounds of k_bigseg.hip
    u64 hugeRows;        // their rows
    u64 bigClusters;     // clusters above CL_BIG rows, left to k_cluster_big
    u64 bigTotal;        // segments > 64 rows of the piles already refined (pile mode: bigCount restarts per pile)
    u64 pad[5];
};

// the tabulated rank queries: one u64 per eBWT row (layout: bfq_rank.h)
struct RankIndex {
    u64 *lfq;               // [n]
    u64 n;
};

struct bfq_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copyStream = nullptr;   // device -> host copies that overlap the tail of the inversion
    bfq_params P;
    BfqEnv env;                         // the BFQ_* environment as bfq_create() / bfq_set_params() found it
    std::string err;

    // workspace arena (bump allocator, reset per top-level call)
    char *ws = nullptr;
    size_t wsCap = 0, wsTop = 0, wsPeak = 0;
    void reserve(size_t bytes);
    void wsFree();
    // an arena being allocated on a helper thread while the caller uploads its input (a one-shot tool's 86 GB can land on
    // memory the driver has not cleared yet: 2 s instead of 0); reserve() joins it and takes it over when it is large enough
    void reserveBegin(size_t bytes);
    void reserveJoin(bool adopt, size_t need);
    std::thread *wsThread = nullptr;
    char *wsPend = nullptr;
    size_t wsPendBytes = 0;
    double wsPendSecs = 0;
    size_t wsVmmChunk = 0;              // != 0: the arena is VMM chunks of this size mapped into one range (experiment)
    std::vector<void *> wsHandles;
    void dropWorkspace();           // frees the arena now (one-shot tools: lets the driver scrub it while outputs are written)
    void *allocBytes(size_t bytes);
    template <class T> T *alloc(size_t count) { return (T *)allocBytes(count * sizeof(T)); }
    size_t mark() const { return wsTop; }
    void release(size_t m) { wsTop = m; }

    DevCounters *d_cnt = nullptr;   // device
    DevCounters h_cnt;              // host copy
    double *d_powtab = nullptr;     // 256 doubles: pow(10,-(q-33)/10) by host libm
    double *d_qthr = nullptr;       // thresholds for round(-10*log10(x))
    int qthrLo = 0, qthrN = 0;

    // last eBWT (device, inside ws)
    u8 *d_bwt = nullptr; u8 *d_qual = nullptr; u16 *d_lcp = nullptr;
    u32 *d_gcnt = nullptr;          // symbol counts per 256-row group, written by k_emit_bwt
    int gcntTerm = -1;              // terminator byte those counts were taken with
    u64 n = 0, N = 0;
    bool piles = false;             // step 1 runs pile by pile (k_piles.hip): set by the reservation of the current call
    bool capped = false;            // the whole path in position mode, one two-symbol pile at a time (workspace cap): set likewise
    u64 cappedPileRows = 0;         // ... and the rows of the largest pile its arena has room for
    size_t wsLimit() const { return env.wsCap ? (size_t)env.wsCap : (size_t)(P.ws_cap_mib > 0 ? P.ws_cap_mib : 0) << 20; }
    bool keepRecs = false;          // step 1 leaves the packed text and the sorted records' (w1, w2) words in the arena (position mode)
    const u64 *d_w12 = nullptr, *d_text3 = nullptr;
    size_t keepMark = 0;

    // profiling
    bool profOn = true;
    std::vector<hipEvent_t> evPool;
    size_t evUsed = 0;
    std::vector<ProfRec> recs;
    double profMs[K_NUM];
    u64 profLaunches[K_NUM];
    double profBytes[K_NUM];
    int profTraceId = -1;               // per-launch times of this kernel id are kept too (bfq_prof_trace), newest last
    std::vector<float> profTrace;
    void profBegin(int id, double bytes);
    void profEnd();
    void profCollect();   // after a stream synchronise

    void sync();
    void fetchCounters();
    void zeroCounters();

    // host <-> device transfers (bfq_io.hip): per-worker stream + two pinned staging buffers
    struct IoWorker { hipStream_t stream = nullptr; char *stage[2] = {nullptr, nullptr}; hipEvent_t done[2] = {nullptr, nullptr}; };
    IoWorker io[BFQ_IO_MAX_WORKERS];
    int ioWorkers = 0;
    void ioInit(int want);          // at least min(want, the thread budget) staging workers exist afterwards
    void ioFree();
    struct BfqWriter *writer = nullptr;  // background device -> host / file writes (bfq_write_async, bfq_io.hip)
    size_t writeHint = 0;               // bytes the caller is going to queue in all: sizes the writer pool when it is created

    // one-shot tools (the *_fd entry points): the eBWT and its qualities live outside the arena (in the text buffer, whose
    // FASTQ text is dead once the reads are gathered), so that the arena can be freed while they are still being written;
    // onRows(start, rows) is called whenever rows [start, start + rows) of the eBWT / QS / LCP are final (pile by pile)
    u8 *extBwt = nullptr, *extQual = nullptr;
    bool lcpScratch = false;            // pile mode: nobody reads the LCP (gsufsort): one pile's worth of scratch instead of 2 n bytes
    std::function<void(u64, u64)> onRows;

    // device copy of the FASTQ text of the current call (outside the arena: its record count sizes the arena);
    // kept between calls, grown when a larger text arrives
    u8 *d_text = nullptr;
    size_t textCap = 0;
    u8 *textBuf(size_t bytes);
    u64 residentLen = 0;            // global mode: length of the block text bfq_glob_begin left in d_text
    bool residentValid = false;
    int residentParts = 0;          // ... and where its parts start (entry residentParts = residentLen)
    u64 residentPstart[BFQ_MAX_PARTS + 1] = {0, 0, 0, 0, 0};
    u64 globN = 0;                  // global mode: the two-symbol pile sizes bfq_glob_pile_counts found for a text of globN rows
    u64 globCounts[36] = {0};
};
// a host-side operand of a transfer: memory, or an open file at an offset (the front-ends' files)
struct HostRef {
    void *ptr = nullptr; int fd = -1; u64 off = 0;
    bfq_outmap *om = nullptr;       // ptr lies `off` bytes into this output mapping: pages are populated before they are written
    static HostRef mem(const void *p) { HostRef h; h.ptr = const_cast<void *>(p); return h; }
    static HostRef file(int fd, u64 off = 0) { HostRef h; h.fd = fd; h.off = off; return h; }
    bool null() const { return !ptr && fd < 0; }
};
void bfq_upload(bfq_ctx *c, void *d_dst, HostRef src, size_t len);
void bfq_download(bfq_ctx *c, HostRef dst, const void *d_src, size_t len);
void bfq_upload(bfq_ctx *c, void *d_dst, const void *h_src, size_t len);
void bfq_download(bfq_ctx *c, void *h_dst, const void *d_src, size_t len);
// background writes: the bytes [d_src, d_src + len) as they are once the work queued so far on the context's stream has
// finished go to dst (memory -- e.g. an output mapping -- or a file offset) through the writer's own staging workers; the
// call returns at once.  bfq_write_wait(): all of them have arrived (throws what went wrong).  d_src must stay valid till then.
void bfq_write_async(bfq_ctx *c, HostRef dst, const void *d_src, size_t len);
void bfq_write_wait(bfq_ctx *c);
// upload on a helper thread while the caller goes on launching kernels; join() waits and rethrows
struct BfqAsyncUpload;
BfqAsyncUpload *bfq_upload_begin(bfq_ctx *c, void *d_dst, HostRef src, size_t len);
void bfq_upload_join(BfqAsyncUpload *u);

#define KLAUNCH(ctx, kid, bytes, kernel, grid, block, ...)                                     \
    do {                                                                                       \
        (ctx)->profBegin((kid), (double)(bytes));                                              \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (ctx)->stream, __VA_ARGS__);    \
        (ctx)->profEnd();                                                                      \
        HIP_CHECK(hipGetLastError());                                                          \
    } while (0)

static inline u64 ceil_div(u64 a, u64 b) { return (a + b - 1) / b; }
// HIP caps gridDim.x * blockDim.x below 2^32: every kernel is launched on at most
// BFQ_MAX_GRID workgroups and strides over its work items.
#define BFQ_MAX_GRID (1u << 19)
static inline unsigned bfq_grid(u64 items, u64 perBlock)
{
    u64 b = ceil_div(items, perBlock);
    return (unsigned)(b < 1 ? 1 : (b > BFQ_MAX_GRID ? BFQ_MAX_GRID : b));
}

// the suffix records being sorted: three 32-bit arrays (layout: bfq_common.h)
struct SortRec { u32 *w0; u64 *w12; };   // w12 = w2 << 32 | w1

// ---- stage entry points (each in its own .hip file) --------------------------------
// scan: out[i] = sum(in[0..i)) ; T in {u8,u32,u64}; total (device u64) optional
void bfq_exscan_u8(bfq_ctx *c, const u8 *in, u64 *out, u64 n, u64 *d_total);
void bfq_exscan_u32(bfq_ctx *c, const u32 *in, u64 *out, u64 n, u64 *d_total);
void bfq_exscan_u64(bfq_ctx *c, const u64 *in, u64 *out, u64 n, u64 *d_total);

u64 *bfq_line_index(bfq_ctx *c, const u8 *d_buf, u64 len, u64 *nlines);   // k_fastq.hip: positions of the line ends (arena)
u64 bfq_fastq_count_lines(bfq_ctx *c, const u8 *d_buf, u64 len);          // k_fastq.hip: newlines + 1
// stream codec (k_codec.hip)
u64 bfq_codec_bound(u64 n);
u64 bfq_codec_workspace(u64 n);
u64 bfq_codec_raw_len(const u8 *h_in, u64 len);                    // all members of a file
u64 bfq_codec_member_len(const u8 *h_in, u64 len);                 // bytes of the first member
u64 bfq_codec_compress_device(bfq_ctx *c, const u8 *d_in, u64 n, u8 *d_out, u64 cap);
u64 bfq_codec_decompress_device(bfq_ctx *c, const u8 *h_in, const u8 *d_in, u64 len, u8 *d_out, u64 cap);

// step 1 pieces
void bfq_build_text(bfq_ctx *c, const u8 *d_bases, const u8 *d_quals, const u64 *d_roff, u64 N, u64 n,
                    u8 *T8, u8 *Q8, u64 *text3, u64 nwords);
// rows per radix workgroup (k_radix.hip); k_build_keys works on the same blocks to leave the first pass's digit counts
#define BFQ_RS_BLOCK_ELEMS 98304                        // rows per radix block (and per text block of the pile counts) for large inputs
#define BFQ_RS_TILE 3072                                // rows per scatter tile
// rows per radix block of an n-row sort: a workgroup runs through its block tile by tile, so small inputs get smaller
// blocks -- at least ~4 rounds of workgroups over the device instead of one round and a tail (1 M x 100 bp: 1028 blocks of
// 98 304 rows on 1024 workgroup slots took two rounds)
static inline u64 bfq_radix_block_elems(u64 n)
{
    u64 tpb = BFQ_RS_BLOCK_ELEMS / BFQ_RS_TILE;
    while (tpb > 1 && n / (BFQ_RS_TILE * tpb) < 4096) tpb >>= 1;   // more, smaller blocks change nothing measurable (30 M x 150: 32, 8 or 4 tiles per block alike)
    return BFQ_RS_TILE * tpb;
}
void bfq_build_keys(bfq_ctx *c, const u8 *T8, const u8 *Q8, const u64 *text3, u64 n, SortRec out, u32 *hist0);   // hist0: [256][ceil(n / bfq_radix_block_elems(n))]
// LSD radix sort of the records on their 48-bit key; result ends in A
struct RadixText { const u8 *T8, *Q8; const u64 *text3; };   // the terminated text the first pass makes its records from
void bfq_key_hist(bfq_ctx *c, const u64 *text3, u64 n, u32 *hist0);                                   // k_text.hip
SortRec bfq_radix_sort(bfq_ctx *c, SortRec in, SortRec tmp, u64 n, int passes = BFQ_KEY_PASSES, const u32 *hist0 = nullptr, const RadixText *fromText = nullptr);   // returns the buffer holding the result (in: even passes, tmp: odd); fewer passes = low digits only; hist0: pass-0 counts already made
// tie refinement: sorts vals inside equal-key segments by the remaining suffix, fills lcp
void bfq_refine(bfq_ctx *c, SortRec rec, const u64 *text3, u64 n, u16 *lcp, bfq_stats *st);
// segments above BFQ_HUGE_SEG rows (listed by k_refine_big): whole-device radix rounds on the following symbols
#define BFQ_HUGE_SEG 2048
void bfq_refine_huge(bfq_ctx *c, SortRec rec, const u64 *text3, u64 n, u16 *lcp, const u64 *hugeStart, const u64 *hugeLen);
void bfq_emit_bwt(bfq_ctx *c, SortRec rec, u64 n, int termOut, u8 *bwt, u8 *qs, u32 *gcnt);   // gcnt: [6][n/256+1] symbol counts
// whole step 1 on device-resident reads; leaves c->d_bwt/d_qual/d_lcp
void bfq_step1_device(bfq_ctx *c, const u8 *d_bases, const u8 *d_quals, const u64 *d_roff, u64 N, u64 total,
                      int termOut, bfq_stats *st);
// the same, one first-symbol pile at a time (k_piles.hip); workspace bound for pile records of at most `cap` rows
struct PileText { u8 *T8, *Q8; u64 *text3; };
void bfq_step1_piles(bfq_ctx *c, const u8 *d_bases, const u8 *d_quals, const u64 *d_roff, u64 N, u64 total, int termOut, bfq_stats *st,
                     const PileText *pre = nullptr, u64 capTarget = 0);
size_t bfq_ws_need_piles(u64 n, u64 N, u64 cap, u64 extra, bool lean = false);
// one pile on its own (global mode): pile-local eBWT / QS / LCP + the sorted records' (w1, w2) words, all in the arena
struct PileRows { u64 m; u8 *bwt, *qs; u16 *lcp; const u64 *w12; };
u64 bfq_run_one_pile(bfq_ctx *c, const u8 *T8, const u8 *Q8, const u64 *text3, u64 n, u32 s, u32 s2, int termOut, PileRows *out);
void bfq_pile_pair_counts(bfq_ctx *c, const u8 *T8, u64 n, u64 *counts36);
void bfq_pack_text(bfq_ctx *c, const u8 *T8, u64 n, u64 *text3, u64 nwords);

// steps 2-4 pieces
RankIndex bfq_rank_build(bfq_ctx *c, const u8 *bwt, const u8 *qs, u64 n, int term, const u32 *gcnt = nullptr);
void bfq_lcp_flags(bfq_ctx *c, const u16 *lcp, u64 n, int K, u8 *in);
// LCP array from the eBWT alone (k_bfs.hip): lcp has n + 1 entries
void bfq_lcp_from_bwt(bfq_ctx *c, const u8 *bwt, u64 n, u64 N, int term, u16 *lcp, u32 *gcntOut = nullptr,   // gcntOut: [6][n/256+1] symbol counts, kept
                      const u64 *rankGiven = nullptr, u64 ringEntries = 0);
// pm != nullptr: position mode -- no LF table (R.lfq may be null): edits go to the output line streams at the text position
// each row's sort record carries (k_cluster.hip)
struct ClusterPos { const u64 *w12; const u64 *text3; u8 *outSym, *outQual; int B; };
// rm != nullptr: rank mode -- no LF table either: LF from the rank blocks, qualities edited in place (qual == the array the
// statistics are read from), replaced bases into repl[] (k_compact.hip)
struct ClusterRank { const u64 *rankBlk; u64 F[6]; u8 *qual, *repl; };
void bfq_clusters(bfq_ctx *c, const RankIndex &R, const u8 *bwt, const u8 *qual, const u8 *in, u64 n, const ClusterPos *pm = nullptr,
                  const ClusterRank *rm = nullptr);
// steps 2-4 on a given eBWT without the LF table (k_compact.hip): rank blocks + qualities in place + replacement array
#define BFQ_COMPACT_LCP_WIN (256ull << 20)                // rows of the LCP file in flight (one upload by all staging workers)
size_t bfq_ws_need_compact(bfq_ctx *c, u64 n, u64 N, u64 extra, bool haveLcp, size_t resident);   // resident: the caller's eBWT + qualities
void bfq_steps234_compact(bfq_ctx *c, const u8 *bwt, u8 *qual, HostRef lcp, int lcp_bytes, u64 n, u64 N, u64 *d_roff, u32 *lens,
                          u8 **ob, u8 **oq, u64 extra, size_t resident);
// LF walks: lengths only, then emission at given offsets
void bfq_invert_count(bfq_ctx *c, const RankIndex &R, u64 N, u32 *lens);
void bfq_fixed_offsets(bfq_ctx *c, u64 N, u64 L, u64 *d_roff);   // d_roff[i] = i * L, i <= N
// reads [first, first + count) (count ~0: all from first); lines: outputs are line streams (read i at roff[i] + i, then '\n')
void bfq_invert(bfq_ctx *c, const RankIndex &R, u64 N, const u64 *d_roff, int B, u8 *out_bases, u8 *out_quals, u64 first = 0,
                u64 count = ~0ull, bool lines = false);
bool bfq_is_pinned(const void *p);

void bfq_synth_launch(bfq_ctx *c, const bfq_synth *s, u8 *d_bases, u8 *d_quals, u64 *d_roff);

// FASTQ text on the device (k_fastq.hip)
struct DevFastq { u64 N, total; void *rec; u64 *roff; u8 *bases, *quals; u64 *lineEnd; };
void bfq_fastq_parse(bfq_ctx *c, const u8 *d_fastq, u64 len, DevFastq *fq);
u64 bfq_fastq_format(bfq_ctx *c, const u8 *d_bases, const u8 *d_quals, const u64 *d_roff, u64 N, int mode, const u8 *d_hdr,
                     u64 hdrLen, const DevFastq *fq, u8 **d_out, u64 **recOffOut = nullptr, bool lines = false);
void bfq_fastq_hdr_stream(bfq_ctx *c, u64 N, const u8 *d_fastq, const DevFastq *fq, u8 **d_hdr, u64 *hdrLen, u64 **hOffOut = nullptr);
