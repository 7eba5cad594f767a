// bfq_api.hip -- C-ABI of libbfqhip.so (include/bfqzip_hip.h) and the host-side
// orchestration of the hot path on one GPU:
//   step 1  reads -> packed text -> (key,payload) pairs -> LSD radix sort -> tie
//           refinement (+LCP) -> eBWT / permuted qualities
//   step 2-3 rank structure -> LCP flags -> clusters (smoothing, replacements)
//   step 4  LF inversion -> reads
// Everything runs on the context's stream inside one device workspace.
#include <math.h>
#include <time.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <errno.h>
#include <algorithm>
#include <new>
#include "bfq_internal.h"
#include <atomic>
extern std::atomic<bool> g_bfqHipStarted;                     // bfq_host.cpp
#include "bfq_synth.h"
#include "bfq_device.h"
#include "bfq_rank.h"

const char *const BFQ_KERNEL_NAMES[K_NUM] = {
    "k_text_from_reads", "k_pack3", "k_build_keys", "k_radix_hist", "k_scan", "k_radix_scatter", "k_huge_round",
    "k_cluster_big", "k_refine_chunk", "k_refine_big", "k_emit_bwt", "k_lf_count", "k_lf_build", "k_lcp_flags",
    "k_cluster", "k_invert_count", "k_invert", "k_synth", "k_fastq", "k_bfs", "k_codec", "misc"};

static thread_local std::string g_createErr;

// ---------------------------------------------------------------- context plumbing
// The arena can also be built from physically contiguous chunks mapped into one address range (HIP's virtual memory
// management: BFQ_WS_VMM=<chunk MiB>): what the 512 write streams of a radix pass cost depends on how the range is backed
// (placement experiments).
void bfq_ctx::reserveBegin(size_t bytes)
{
    bytes = (bytes + 0xFFFFF) & ~(size_t)0xFFFFF;
    if (wsThread || bytes <= wsCap || env.wsVmmMib || env.wsContig || (wsLimit() && bytes > wsLimit())) return;
    wsPendBytes = bytes; wsPend = nullptr;
    const int dev = device;
    wsThread = new std::thread([this, dev, bytes] {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        void *p = nullptr;
        if (hipSetDevice(dev) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        wsPend = (char *)p;
        wsPendSecs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    });
}
// adopt: the pending arena replaces the current one when it holds `need` bytes (and the current one does not)
void bfq_ctx::reserveJoin(bool adopt, size_t need)
{
    if (!wsThread) return;
    wsThread->join();
    delete wsThread;
    wsThread = nullptr;
    if (!wsPend) return;
    if (adopt && need <= wsPendBytes && need > wsCap && !(wsLimit() && wsPendBytes > wsLimit())) {
        wsFree();
        ws = wsPend; wsCap = wsPendBytes; wsTop = 0;
        if (bfq_env().trace) fprintf(stderr, "[bfq] workspace %.1f GiB: hipMalloc %.3f s, beside the upload\n", wsCap / 1073741824.0, wsPendSecs);
    } else (void)hipFree(wsPend);
    wsPend = nullptr; wsPendBytes = 0;
}
void bfq_ctx::wsFree()
{
    if (!ws) return;
    (void)hipStreamSynchronize(stream);
    if (wsVmmChunk) {
        (void)hipMemUnmap(ws, wsCap);
        for (auto h : wsHandles) (void)hipMemRelease((hipMemGenericAllocationHandle_t)h);
        wsHandles.clear();
        (void)hipMemAddressFree(ws, wsCap);
        wsVmmChunk = 0;
    } else (void)hipFree(ws);
    ws = nullptr; wsCap = 0; wsTop = 0;
}
static hipError_t ws_alloc_vmm(bfq_ctx *c, size_t bytes, size_t chunk, char **out, size_t *got)
{
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess) return e;
    if (chunk < gran) chunk = gran;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t total = (bytes + chunk - 1) / chunk * chunk;
    void *va = nullptr;
    e = hipMemAddressReserve(&va, total, chunk, nullptr, 0);
    if (e != hipSuccess) return e;
    std::vector<void *> hs;
    for (size_t o = 0; o < total && e == hipSuccess; o += chunk) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, chunk, &prop, 0);
        if (e == hipSuccess) { hs.push_back((void *)h); e = hipMemMap((char *)va + o, chunk, 0, h, 0); }
    }
    if (e == hipSuccess) {
        hipMemAccessDesc d;
        memset(&d, 0, sizeof d);
        d.location.type = hipMemLocationTypeDevice; d.location.id = c->device; d.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, total, &d, 1);
    }
    if (e != hipSuccess) {
        (void)hipMemUnmap(va, total);
        for (auto h : hs) (void)hipMemRelease((hipMemGenericAllocationHandle_t)h);
        (void)hipMemAddressFree(va, total);
        return e;
    }
    c->wsHandles = hs; c->wsVmmChunk = chunk;
    *out = (char *)va; *got = total;
    return hipSuccess;
}

void bfq_ctx::reserve(size_t bytes)
{
    bytes = (bytes + 0xFFFFF) & ~(size_t)0xFFFFF;
    reserveJoin(true, bytes);
    if (wsLimit() && bytes > wsLimit()) {
        char b[200];
        snprintf(b, sizeof b, "device workspace of %.1f GiB is above the cap of %.1f GiB (bfq_params.ws_cap_mib / BFQ_WS_CAP)", bytes / 1073741824.0, wsLimit() / 1073741824.0);
        throw BfqError{BFQ_E_NOMEM, b};
    }
    if (bytes > wsCap || (wsLimit() && wsCap > wsLimit())) {    // (an arena from before the cap was set goes back)
        wsFree();
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        hipError_t e = hipErrorOutOfMemory;
        bool contig = false;
        if (env.wsVmmMib) {
            size_t got = 0;
            e = ws_alloc_vmm(this, bytes, (size_t)env.wsVmmMib << 20, &ws, &got);
            if (e == hipSuccess) bytes = got; else { (void)hipGetLastError(); ws = nullptr; }
            if (bfq_env().trace) fprintf(stderr, "[bfq] workspace: VMM chunks of %d MiB %s\n", env.wsVmmMib, e == hipSuccess ? "mapped" : "refused");
        }
        if (e != hipSuccess && env.wsContig) { e = hipExtMallocWithFlags((void **)&ws, bytes, hipDeviceMallocContiguous); contig = e == hipSuccess; if (!contig) (void)hipGetLastError(); }
        if (e != hipSuccess) e = hipMalloc((void **)&ws, bytes);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (bfq_env().trace && env.wsContig) fprintf(stderr, "[bfq] workspace: contiguous allocation %s\n", contig ? "granted" : "refused");
        if (bfq_env().trace) fprintf(stderr, "[bfq] workspace %.1f GiB: hipMalloc %.3f s\n", bytes / 1073741824.0, (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
        if (e != hipSuccess) {
            ws = nullptr;
            char b[160];
            snprintf(b, sizeof b, "device workspace of %zu bytes: %s", bytes, hipGetErrorString(e));
            (void)hipGetLastError();
            throw BfqError{BFQ_E_NOMEM, b};
        }
        wsCap = bytes;
    }
    wsTop = 0;
    d_bwt = d_qual = nullptr; d_lcp = nullptr; d_gcnt = nullptr; n = N = 0;
}
void bfq_ctx::dropWorkspace()
{
    wsFree();
    d_bwt = d_qual = nullptr; d_lcp = nullptr; d_gcnt = nullptr;
}
void *bfq_ctx::allocBytes(size_t bytes)
{
    size_t a = (wsTop + 255) & ~(size_t)255;
    if (a + bytes > wsCap) {
        char b[160];
        snprintf(b, sizeof b, "workspace exhausted: need %zu more bytes at %zu of %zu", bytes, a, wsCap);
        throw BfqError{BFQ_E_NOMEM, b};
    }
    wsTop = a + bytes;
    if (wsTop > wsPeak) wsPeak = wsTop;
    return ws + a;
}
void bfq_ctx::profBegin(int id, double bytes)
{
    if (!profOn) return;
    if (evUsed + 2 > evPool.size()) {
        size_t old = evPool.size();
        evPool.resize(old + 256);
        for (size_t i = old; i < evPool.size(); i++) HIP_CHECK(hipEventCreate(&evPool[i]));
    }
    ProfRec r{id, evPool[evUsed], evPool[evUsed + 1], bytes};
    evUsed += 2;
    HIP_CHECK(hipEventRecord(r.a, stream));
    recs.push_back(r);
}
void bfq_ctx::profEnd()
{
    if (!profOn) return;
    HIP_CHECK(hipEventRecord(recs.back().b, stream));
}
void bfq_ctx::profCollect()
{
    for (auto &r : recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            profMs[r.id] += ms; profLaunches[r.id]++; profBytes[r.id] += r.bytes;
            if (r.id == profTraceId && profTrace.size() < (1u << 20)) profTrace.push_back(ms);
        }
    }
    recs.clear();
    evUsed = 0;
}
void bfq_ctx::sync() { HIP_CHECK(hipStreamSynchronize(stream)); }
void bfq_ctx::zeroCounters() { HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(DevCounters), stream)); }
void bfq_ctx::fetchCounters()
{
    HIP_CHECK(hipMemcpyAsync(&h_cnt, d_cnt, sizeof(DevCounters), hipMemcpyDeviceToHost, stream));
    sync();
}

// round(-10*log10(x)) exactly as the host libm evaluates it (bfq_int.cpp:370)
static int host_q(double x) { return (int)round(-10 * log10(x)); }

extern "C" void bfq_default_params(bfq_params *p)
{
    memset(p, 0, sizeof(*p));
    p->K = 16; p->m = 2; p->v = '>'; p->f = 40; p->t = 20; p->term = '#'; p->M = 2; p->B = 0; p->ext = 0; p->piles = 0;
}
extern "C" const char *bfq_create_error(void) { return g_createErr.c_str(); }
extern "C" const char *bfq_version(void) { return "bfqzip_amd 0.1 (gfx950)"; }
extern "C" int bfq_device_count(void)
{
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nd;
}

// the GPU a one-shot tool runs on (include/bfqzip_hip.h): lease by PCI bus id, so that processes with different
// HIP_VISIBLE_DEVICES still agree on which lock file stands for which GPU
extern "C" int bfq_pick_device(char *info, int info_cap)
{
    const BfqEnv &E = bfq_env();
    int real = 0;
    if (hipGetDeviceCount(&real) != hipSuccess || real < 1) { (void)hipGetLastError(); return BFQ_E_HIP; }
    const int slots = E.fakeDevices > 0 ? E.fakeDevices : real;
    if (E.device >= slots) return BFQ_E_ARG;
    if (!E.lease) {
        const int d = E.device >= 0 ? E.device % real : 0;
        if (info && info_cap > 0) snprintf(info, info_cap, "device %d (no lease)", d);
        return d;
    }
    std::vector<std::string> ids(slots);
    std::vector<const char *> idp(slots);
    for (int k = 0; k < slots; k++) {
        char bus[64] = {0};
        if (E.fakeDevices > 0) ids[k] = "fake" + std::to_string(k) + "of" + std::to_string(slots);
        else if (hipDeviceGetPCIBusId(bus, sizeof bus, k) == hipSuccess && bus[0]) ids[k] = std::string("gpu-") + bus;
        else { (void)hipGetLastError(); ids[k] = "gpu-index" + std::to_string(k); }
        idp[k] = ids[k].c_str();
    }
    char path[256] = {0};
    double waited = 0;
    const int slot = bfq_device_lease(slots, idp.data(), E.device, path, sizeof path, &waited);
    if (slot < 0) return slot;
    if (info && info_cap > 0) snprintf(info, info_cap, "device %d lease %s waited %.3f s", slot % real, path, waited);
    return slot % real;
}

extern "C" bfq_ctx *bfq_create(int device, const bfq_params *p)
{
    bfq_ctx *c = nullptr;
    try {
        int nd = 0;
        if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0)
            throw BfqError{BFQ_E_HIP, "no HIP device available (libbfqhip.so has no CPU fallback)"};
        if (device < 0 || device >= nd) throw BfqError{BFQ_E_ARG, "device index out of range"};
        c = new bfq_ctx();
        c->device = device;
        if (p) c->P = *p; else bfq_default_params(&c->P);
        c->env = bfq_env_read();
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIP_CHECK(hipMalloc((void **)&c->d_cnt, sizeof(DevCounters)));
        memset(c->profMs, 0, sizeof c->profMs); memset(c->profLaunches, 0, sizeof c->profLaunches);
        memset(c->profBytes, 0, sizeof c->profBytes);
        g_bfqHipStarted = true;                                    // (bfq_host.cpp: the output files' populate helpers may map pages now)
        // tables for M=1, computed with the host libm the reference itself would use
        double pw[256];
        for (int q = 0; q < 256; q++) pw[q] = pow(10, -((double)(signed char)q - 33) / 10);
        c->qthrLo = -170; c->qthrN = 311;
        std::vector<double> thr(c->qthrN);
        for (int k = 0; k < c->qthrN; k++) {
            int q = c->qthrLo + k;
            double xlo = 1e-30, xhi = 1e30;           // host_q(xlo) > q, host_q(xhi) <= q
            u64 lo, hi;
            memcpy(&lo, &xlo, 8); memcpy(&hi, &xhi, 8);
            while (hi - lo > 1) {
                u64 mid = lo + (hi - lo) / 2;
                double xm; memcpy(&xm, &mid, 8);
                if (host_q(xm) <= q) hi = mid; else lo = mid;
            }
            memcpy(&thr[k], &hi, 8);
        }
        HIP_CHECK(hipMalloc((void **)&c->d_powtab, sizeof pw));
        HIP_CHECK(hipMalloc((void **)&c->d_qthr, sizeof(double) * c->qthrN));
        HIP_CHECK(hipMemcpy(c->d_powtab, pw, sizeof pw, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(c->d_qthr, thr.data(), sizeof(double) * c->qthrN, hipMemcpyHostToDevice));
        return c;
    } catch (const BfqError &e) {
        g_createErr = e.msg;
    } catch (const std::exception &e) {
        g_createErr = e.what();
    }
    delete c;
    return nullptr;
}

extern "C" void bfq_destroy(bfq_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto e : c->evPool) (void)hipEventDestroy(e);
    c->ioFree();
    if (c->d_text) (void)hipFree(c->d_text);
    if (bfq_env().trace && c->wsCap) fprintf(stderr, "[bfq] workspace: %.1f GiB reserved, %.1f GiB used at the peak\n", c->wsCap / 1073741824.0, c->wsPeak / 1073741824.0);
    c->wsFree();
    if (c->d_cnt) (void)hipFree(c->d_cnt);
    if (c->d_powtab) (void)hipFree(c->d_powtab);
    if (c->d_qthr) (void)hipFree(c->d_qthr);
    if (c->copyStream) (void)hipStreamDestroy(c->copyStream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
extern "C" int bfq_set_params(bfq_ctx *c, const bfq_params *p)
{
    if (!c || !p) return BFQ_E_ARG;
    c->P = *p;
    c->env = bfq_env_read();
    return BFQ_OK;
}
extern "C" const char *bfq_last_error(bfq_ctx *c) { return c ? c->err.c_str() : "null context"; }
extern "C" void *bfq_stream(bfq_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" uint64_t bfq_workspace_bytes(bfq_ctx *c) { return c ? c->wsCap : 0; }
extern "C" int bfq_prof_enable(bfq_ctx *c, int on) { if (!c) return BFQ_E_ARG; c->profOn = on != 0; return BFQ_OK; }
extern "C" void bfq_prof_reset(bfq_ctx *c)
{
    if (!c) return;
    memset(c->profMs, 0, sizeof c->profMs); memset(c->profLaunches, 0, sizeof c->profLaunches);
    memset(c->profBytes, 0, sizeof c->profBytes);
    c->profTrace.clear();
}
// per-launch durations of ONE kernel (by its bfq_prof_get index; -1: none) in launch order since the last bfq_prof_reset():
// bfq_prof_trace_select() chooses it, bfq_prof_trace() copies up to cap values and returns how many there are
extern "C" int bfq_prof_trace_select(bfq_ctx *c, int idx)
{
    if (!c || idx < -1 || idx >= K_NUM) return BFQ_E_ARG;
    c->profTraceId = idx; c->profTrace.clear();
    return BFQ_OK;
}
extern "C" int64_t bfq_prof_trace(bfq_ctx *c, float *ms, uint64_t cap)
{
    if (!c) return BFQ_E_ARG;
    const size_t k = std::min<size_t>(cap, c->profTrace.size());
    if (ms && k) memcpy(ms, c->profTrace.data(), k * sizeof(float));
    return (int64_t)c->profTrace.size();
}
extern "C" int bfq_prof_count(bfq_ctx *c) { (void)c; return K_NUM; }
extern "C" int bfq_prof_get(bfq_ctx *c, int idx, char *name, int cap, double *ms, uint64_t *launches, double *bytes)
{
    if (!c || idx < 0 || idx >= K_NUM) return BFQ_E_ARG;
    if (name && cap > 0) { strncpy(name, BFQ_KERNEL_NAMES[idx], cap - 1); name[cap - 1] = 0; }
    if (ms) *ms = c->profMs[idx];
    if (launches) *launches = c->profLaunches[idx];
    if (bytes) *bytes = c->profBytes[idx];
    return BFQ_OK;
}

// run `body` with the usual prologue/epilogue; maps exceptions to error codes
template <class F> static int guarded(bfq_ctx *c, F body)
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

static void check_counters(bfq_ctx *c)
{
    const DevCounters &h = c->h_cnt;
    if (h.errSymbol) throw BfqError{BFQ_E_SYMBOL, "symbol outside {A,C,G,T,N,terminator}"};
    if (h.errTooLong) throw BfqError{BFQ_E_TOO_LONG, "read longer than BFQ_MAX_READ_LEN"};
    if (h.errInvert) throw BfqError{BFQ_E_NOT_EBWT, "LF walk did not close: not an eBWT of a read collection"};
    if (h.mismatch) throw BfqError{BFQ_E_NOT_EBWT, "eBWT is not in #_i<#_j<A<C<G<N<T suffix order"};
    if (h.errFreq3) throw BfqError{BFQ_E_FREQ3, "three frequent symbols in a cluster (bfq_int.cpp:505 assert); raise -f"};
}
static void fill_stats(bfq_ctx *c, bfq_stats *st)
{
    if (!st) return;
    const u64 *s = c->h_cnt.stats;
    st->num_clust = s[0]; st->num_clust_discarded = s[1]; st->num_clust_amb_discarded = s[2];
    st->num_clust_mod = s[3]; st->num_clust_alleq = s[4]; st->bases_inside = s[5];
    st->qs_smoothed = s[6]; st->modified = s[7];
    st->n_rows = c->n; st->n_reads = c->N;
    st->n_segments = c->h_cnt.nSegs; st->n_big_segments = c->h_cnt.bigTotal + c->h_cnt.bigCount;
}

// workspace bound for a collection of n rows (see DESIGN.md "HBM layout")
static size_t ws_need(u64 n, u64 N, u64 extra)
{
    const u64 nb = ceil_div(n + 1, bfq_radix_block_elems(n)) + 8;   // small sorts use smaller radix blocks (bfq_radix_block_elems)
    size_t need = 0;
    need += 4 * (n + 256) + 24 * (n / 256 + 2) + 4096;  // bwt, qual, lcp16, symbol counts per group
    need += 8 * bfq_t3_alloc(n / 21 + 3);                           // packed text
    need += 6 * 4 * (n + 256);                          // sort records, ping-pong (2 x 12 B/row)
    need += 256 * nb * 12 + (nb + 4096) * 64;           // radix histograms + scan partials
    need += 16 * (N + 64);                              // offsets / lengths
    need += extra + (64u << 20);
    return need;
}

// the same for steps 2-4 on a given eBWT: eBWT + qualities + reads out (4 n), LCP (2 n), LF table (8 n), flags (n);
// the interval refinement's rank blocks and queue (9 n) live where the LF table and the flags come afterwards
// `extra` (the caller's formatted output) is allocated when the LF table and the cluster tables have gone back to the arena: it
// shares their 10 bytes per row (bfq_int on 30 M x 150: 83.5 GiB were reserved for a peak of 64.4 -- the fewer blocks the
// arena draws, the rarer the wait for memory the previous process has just given back)
static size_t ws_need_given(u64 n, u64 N, u64 extra)
{
    size_t need = 0;
    need += 6 * (n + 256) + 4096;
    const size_t table = 10 * (n + 1024) + 72 * (n / 256 + 2);
    need += table > extra ? table : extra;
    need += 16 * (N + 64) + (n >> 20) * 64 + 4096;
    need += 64u << 20;
    return need;
}

// the capped mode (steps_capped): T8 + Q8 + the two line streams (4 n), packed text, block counts, one pile of `pile` rows
// (pile-local eBWT / QS / LCP 4 B, sort records 24 B, flags and cluster tables ~4 B per row)
static size_t ws_need_capped(u64 n, u64 N, u64 pile, u64 extra)
{
    size_t need = 0;
    need += 4 * (n + 512) + 8 * bfq_t3_alloc(n / 21 + 3);
    need += 2 * 6 * 12 * (n / 32768 + 2) + 16 * (N + 64);
    if (pile) need += 32 * (pile + 256) + 256 * 12 * (ceil_div(pile + 1, bfq_radix_block_elems(pile)) + 8200) + (pile / 32768 + 4096) * 64;
    need += extra + (96u << 20);
    return need;
}

// Workspace for a call that runs step 1 on n rows: in one piece (ws_need) or pile by pile (bfq_params.piles: 1 always,
// 0 = when the one-piece workspace cannot be had, -1 never; the environment variable BFQ_PILES=0/1 overrides).
static void reserve_step1(bfq_ctx *c, u64 n, u64 N, u64 extra, bool allowCapped = true)
{
    int mode = c->P.piles;
    if (c->env.piles) mode = c->env.piles;
    if (mode == 2 && !allowCapped) mode = 1;                     // the caller wants the eBWT arrays: the capped mode has none
    const u64 cap = n / 10 * 3 + (1u << 20);                     // a DNA pile holds about a quarter of the suffixes; larger ones are split again
    c->piles = false; c->capped = false;
    if (mode <= 0) {
        try { c->reserve(ws_need(n, N, extra + c->env.abPad + ((c->env.abSwap || c->env.abOrder[0]) ? 12 * (n + 256) : 0))); return; }
        catch (const BfqError &e) { if (mode < 0 || e.code != BFQ_E_NOMEM) throw; }
    }
    if (mode != 2) {
        try { c->reserve(bfq_ws_need_piles(n, N, cap, extra)); c->piles = true; return; }
        catch (const BfqError &e) { if (e.code != BFQ_E_NOMEM || !c->wsLimit() || !allowCapped) throw; }
    }
    // Below 13 n bytes (a workspace cap): the capped mode.  Its arena holds the terminated text (T8, Q8, packed text), the
    // line streams the edits go to, and ONE two-symbol pile at a time -- as large a pile as the cap leaves room for.
    const size_t base = ws_need_capped(n, N, 0, extra), lim = c->wsLimit();
    u64 rows = !lim ? n / 4 + (1u << 20) : lim > base + (128u << 20) ? (u64)((lim - base - (128u << 20)) / 32) : 0;   // (no cap: bfq_params.piles = 2 asked for the mode)
    if (rows > n / 4 + (1u << 20)) rows = n / 4 + (1u << 20);
    if (lim && rows < n / 64 + 4096) {
        char b[220];
        snprintf(b, sizeof b, "workspace cap of %.1f GiB too small for %llu rows: the capped mode needs %.1f GiB + 32 bytes per row of its largest two-symbol pile",
                 lim / 1073741824.0, (unsigned long long)n, base / 1073741824.0);
        throw BfqError{BFQ_E_NOMEM, b};
    }
    c->reserve(ws_need_capped(n, N, rows, extra));
    c->capped = true; c->cappedPileRows = rows;
}

// ---------------------------------------------------------------- step 1
void bfq_step1_device(bfq_ctx *c, const u8 *d_bases, const u8 *d_quals, const u64 *d_roff, u64 N, u64 total,
                      int termOut, bfq_stats *st)
{
    if (c->piles) { bfq_step1_piles(c, d_bases, d_quals, d_roff, N, total, termOut, st); return; }
    u64 n = total + N;
    if (n >= (1ull << BFQ_POS_BITS)) throw BfqError{BFQ_E_ARG, "collection too large (2^37 rows)"};
    c->n = n; c->N = N;
    c->d_bwt = c->extBwt ? c->extBwt : c->alloc<u8>(n + 64);
    c->d_qual = c->extQual ? c->extQual : c->alloc<u8>(n + 64);
    c->d_lcp = c->alloc<u16>(n + 64);
    c->d_gcnt = c->alloc<u32>(6 * (n / 256 + 1));
    c->gcntTerm = termOut & 0xFF;
    if (!n) return;
    size_t m0 = c->mark();
    u64 nwords = n / BFQ_SYMS_PER_WORD + 3;
    u64 *text3 = c->alloc<u64>(bfq_t3_alloc(nwords));
    SortRec A, B;
    const bool abSwap = c->env.abSwap && !c->keepRecs;  // placement experiment: B below A
    size_t mKeep = 0, mB = 0;
    if (c->env.abOrder[0] && !c->keepRecs) {            // placement experiment: the four arrays in any order (0 A.w12, 1 A.w0, 2 B.w0, 3 B.w12)
        for (int k = 0; k < 4; k++) {
            switch (c->env.abOrder[k]) {
            case '0': A.w12 = c->alloc<u64>(n + 16); break;
            case '1': A.w0 = c->alloc<u32>(n + 16); break;
            case '2': B.w0 = c->alloc<u32>(n + 16); break;
            default: B.w12 = c->alloc<u64>(n + 16); break;
            }
        }
        mKeep = mB = c->mark();
    } else {
    if (abSwap) { B.w0 = c->alloc<u32>(n + 16); B.w12 = c->alloc<u64>(n + 16); }
    A.w12 = c->alloc<u64>(n + 16);
    mKeep = c->mark();                                  // position mode keeps the text and the records' (w1, w2) words
    A.w0 = c->alloc<u32>(n + 16);
    mB = c->mark();
    if (c->env.abPad) (void)c->allocBytes((size_t)c->env.abPad);
    if (!abSwap) { B.w0 = c->alloc<u32>(n + 16); B.w12 = c->alloc<u64>(n + 16); }
    }
    if (bfq_env().trace) fprintf(stderr, "[bfq] sort buffers: A.w12 %p A.w0 %p B.w0 %p B.w12 %p (arena %p)\n", (void *)A.w12, (void *)A.w0, (void *)B.w0, (void *)B.w12, (void *)c->ws);
    // k_build_keys writes the records to B, the byte text lying in A (dead before the sort's first scatter writes there).
    // BFQ_KEY_FUSION=1 (tried in round 3, slower): the records never exist unsorted -- the first pass of the sort makes them
    // from the text on its way (k_radix_scatter<2>; the byte text then lies in B, which that pass does not touch).  It saves
    // 12 B/row written + 9.6 B/row read, but the key of a suffix (window, terminator mask, 40-bit packing) costs the scatter
    // kernel more than the traffic it saves: k_build_keys 17.3 -> 6.4 ms (counts only), pass 0 36.4 -> 66.1 ms, step +17 ms.
    const bool fused = c->env.keyFusion;
    u8 *T8 = fused ? (u8 *)B.w0 : (u8 *)A.w0, *Q8 = fused ? (u8 *)B.w12 : (u8 *)A.w12;
    bfq_build_text(c, d_bases, d_quals, d_roff, N, n, T8, Q8, text3, nwords);
    u32 *hist0 = c->alloc<u32>(256 * ceil_div(n, bfq_radix_block_elems(n)));
    static_assert(BFQ_KEY_PASSES & 1, "an odd number of passes ends in the other buffer");
    if (fused) {
        bfq_key_hist(c, text3, n, hist0);               // the first pass's digit counts
        const RadixText tx{T8, Q8, text3};
        bfq_radix_sort(c, B, A, n, BFQ_KEY_PASSES, hist0, &tx);   // text -> A -> B -> A -> B -> A
    } else {
        bfq_build_keys(c, T8, Q8, text3, n, B, hist0);
        bfq_radix_sort(c, B, A, n, BFQ_KEY_PASSES, hist0);
    }
    c->release(mB);                                     // the big-segment list reuses the B buffers
    bfq_refine(c, A, text3, n, c->d_lcp, st);
    bfq_emit_bwt(c, A, n, termOut, c->d_bwt, c->d_qual, c->d_gcnt);
    if (c->keepRecs) { c->release(mKeep); c->d_w12 = A.w12; c->d_text3 = text3; c->keepMark = m0; }
    else c->release(m0);
    if (c->onRows) c->onRows(0, n);
}

// ---------------------------------------------------------------- steps 2-4
// lens != nullptr: the read lengths are not known (an eBWT given from outside).  Collections of equal-length reads
// (n - N a multiple of N) are first inverted on that assumption -- k_invert checks every walk against its slot, so a
// wrong guess is noticed -- which saves the counting walk; otherwise, or when the guess fails, the lengths are counted
// by LF walks and d_roff (N + 1 entries) receives their offsets; the walks must cover the eBWT.
static void count_lengths(bfq_ctx *c, const RankIndex &R, u64 *d_roff, u32 *lens)
{
    const u64 n = c->n, N = c->N;
    bfq_invert_count(c, R, N, lens);
    bfq_exscan_u32(c, lens, d_roff, N, d_roff + N);
    u64 tot2 = 0;
    HIP_CHECK(hipMemcpyAsync(&tot2, d_roff + N, 8, hipMemcpyDeviceToHost, c->stream));
    c->fetchCounters();
    check_counters(c);
    if (tot2 != n - N) throw BfqError{BFQ_E_NOT_EBWT, "LF walks do not cover the eBWT"};
}
// so != nullptr: the outputs are the line streams OUT.fq.dna / OUT.fq.qs (d_out_bases / d_out_quals hold total + N bytes);
// with pinned host destinations the inversion runs in read-range chunks and every finished chunk is copied to the host on
// a second stream while the next one is walked.
struct StreamOut { u8 *h_dna, *h_qs; bool packed = false; };   // packed: the reads back to back (read i at roff[i]) instead of line streams
static void invert_lines(bfq_ctx *c, const RankIndex &R, const u64 *d_roff, u8 *d_dna, u8 *d_qs, const StreamOut *so)
{
    const u64 N = c->N;
    const bool lines = !so->packed;
    const bool pinned = (!so->h_dna || bfq_is_pinned(so->h_dna)) && (!so->h_qs || bfq_is_pinned(so->h_qs));
    const int C = (pinned && N >= (1u << 16) && !c->env.noOverlap) ? 8 : 1;
    if (C == 1) {
        bfq_invert(c, R, N, d_roff, c->P.B, d_dna, d_qs, 0, ~0ull, lines);
        const u64 sl = (c->n - N) + (lines ? N : 0);
        if (so->h_dna) bfq_download(c, so->h_dna, d_dna, sl);
        if (so->h_qs) bfq_download(c, so->h_qs, d_qs, sl);
        return;
    }
    if (!c->copyStream) HIP_CHECK(hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
    u64 first[9], off[9];
    for (int j = 0; j <= C; j++) first[j] = N * (u64)j / C;
    for (int j = 0; j <= C; j++) HIP_CHECK(hipMemcpyAsync(&off[j], d_roff + first[j], 8, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    hipEvent_t ev[8];
    for (int j = 0; j < C; j++) {
        HIP_CHECK(hipEventCreateWithFlags(&ev[j], hipEventDisableTiming));
        bfq_invert(c, R, N, d_roff, c->P.B, d_dna, d_qs, first[j], first[j + 1] - first[j], lines);
        HIP_CHECK(hipEventRecord(ev[j], c->stream));
        HIP_CHECK(hipStreamWaitEvent(c->copyStream, ev[j], 0));
        const u64 b0 = off[j] + (lines ? first[j] : 0), b1 = off[j + 1] + (lines ? first[j + 1] : 0);
        if (so->h_dna) HIP_CHECK(hipMemcpyAsync(so->h_dna + b0, d_dna + b0, b1 - b0, hipMemcpyDeviceToHost, c->copyStream));
        if (so->h_qs) HIP_CHECK(hipMemcpyAsync(so->h_qs + b0, d_qs + b0, b1 - b0, hipMemcpyDeviceToHost, c->copyStream));
    }
    HIP_CHECK(hipStreamSynchronize(c->copyStream));
    for (int j = 0; j < C; j++) (void)hipEventDestroy(ev[j]);
}
// eBWT-domain output (bfq_fastq_job.compress_streams == 2): the rows of the eBWT as the cluster step left them -- symbol after
// noise reduction, quality after smoothing (binned when B = 1; a constant at the terminator rows, whose quality byte never
// reaches a FASTQ) -- instead of the reads they invert to.  In row order the symbols of a deep collection are runs: they
// code to half the size of the read-order stream, and the compressing side skips the inversion altogether.
struct EbwtOut { u8 *sym, *qual, *patch; u8 *lineDna, *lineQs; };   // lineQs != nullptr: the qualities in READ order as well (a walk)
// patch[r] = the ORIGINAL symbol of a row whose base was replaced, else 0: the walk navigates by the original eBWT
// (invert, bfq_int.cpp:782-790: base = replaced ? BWT_MOD : bwt[j], j = LF(j) of the unchanged BWT).
__global__ __launch_bounds__(256) void k_ebwt_rows(const u64 *__restrict__ lfq, u64 n, u32 term, int B, u8 *__restrict__ sym, u8 *__restrict__ qual,
                                                   u8 *__restrict__ patch)
{
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x) {
        const u64 x = lfq[r];
        const u32 code = lfq_replaced(x) ? lfq_repl(x) : lfq_code(x);
        sym[r] = code ? bfq_code_sym(code) : (u8)term;
        patch[r] = lfq_replaced(x) ? bfq_code_sym(lfq_code(x)) : (u8)0;
        const u32 q = lfq_qual(x);
        qual[r] = lfq_code(x) ? (u8)(B ? bfq_bin8(q) : q) : (u8)'!';
    }
}
static void steps234_device(bfq_ctx *c, u64 *d_roff, u8 *d_out_bases, u8 *d_out_quals, u32 *lens = nullptr, const StreamOut *so = nullptr,
                            const EbwtOut *eo = nullptr)
{
    u64 n = c->n, N = c->N;
    if (!n) return;
    // the symbol counts of step 1's emission (or of the interval refinement) are reused when they belong to this eBWT
    const u32 *gc = (c->d_gcnt && c->gcntTerm == (c->P.term & 0xFF)) ? c->d_gcnt : nullptr;
    RankIndex R = bfq_rank_build(c, c->d_bwt, c->d_qual, n, c->P.term, gc);
    bool guessed = false;
    if (lens) {
        if (N && (n - N) % N == 0 && !c->env.noLengthGuess) { bfq_fixed_offsets(c, N, (n - N) / N, d_roff); guessed = true; }
        else count_lengths(c, R, d_roff, lens);
    }
    u8 *in = c->alloc<u8>(n + 64);
    bfq_lcp_flags(c, c->d_lcp, n, c->P.K, in);
    bfq_clusters(c, R, c->d_bwt, c->d_qual, in, n);
    if (eo) {
        KLAUNCH(c, K_MISC, 10.0 * (double)n, k_ebwt_rows, bfq_grid(n, 256), 256, (const u64 *)R.lfq, n, (u32)(c->P.term & 0xFF), c->P.B, eo->sym, eo->qual, eo->patch);
        if (eo->lineQs) { StreamOut none{nullptr, nullptr}; invert_lines(c, R, d_roff, eo->lineDna, eo->lineQs, &none); }
        return;
    }
    if (so) {
        invert_lines(c, R, d_roff, d_out_bases, d_out_quals, so);
        if (guessed) {
            c->fetchCounters();
            if (c->h_cnt.errInvert) {                              // not all of one length after all: count, then walk again
                HIP_CHECK(hipMemsetAsync(&c->d_cnt->errInvert, 0, sizeof(u64), c->stream));
                count_lengths(c, R, d_roff, lens);
                invert_lines(c, R, d_roff, d_out_bases, d_out_quals, so);
            }
        }
        return;
    }
    bfq_invert(c, R, N, d_roff, c->P.B, d_out_bases, d_out_quals);
    if (guessed) {
        c->fetchCounters();
        if (c->h_cnt.errInvert) {                              // not all of one length after all: count, then walk again
            HIP_CHECK(hipMemsetAsync(&c->d_cnt->errInvert, 0, sizeof(u64), c->stream));
            count_lengths(c, R, d_roff, lens);
            bfq_invert(c, R, N, d_roff, c->P.B, d_out_bases, d_out_quals);
        }
    }
}

// ---- the fused path without LF table and walks ("position mode", device-resident entry point only).
// When step 1 has just run, every row still knows the text position of its suffix (the sort payload), so steps 3-4 need
// neither LF(row) nor an inversion: the outputs start as a copy of the input (qualities binned when B = 1) in line-stream
// layout, k_cluster writes its edits straight to the position each row stands for, and the terminators are dropped
// again at the end.  What it trades: k_lf_build + k_invert (one random 64-byte sector per base) against one random
// byte store per EDITED base or quality (bfq_params / BFQ_POSMODE; measured in DESIGN.md 4).
__global__ __launch_bounds__(256) void k_lines_init(const u8 *__restrict__ bases, const u8 *__restrict__ quals, const u64 *__restrict__ roff, u64 N,
                                                    int B, u8 *__restrict__ dna, u8 *__restrict__ qs)
{
    const u32 sub = threadIdx.x & 15u;
    const u64 ngrp = ((u64)gridDim.x * blockDim.x) >> 4;
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < N; i += ngrp) {
        const u64 b = roff[i], len = roff[i + 1] - b, o = b + i;
        for (u64 k = sub; k < len; k += 16) { dna[o + k] = bases[b + k]; const u32 q = quals[b + k]; qs[o + k] = (u8)(B ? bfq_bin8(q) : q); }
        if (sub == 0) { dna[o + len] = 10; qs[o + len] = 10; }
    }
}
__global__ __launch_bounds__(256) void k_lines_strip(const u8 *__restrict__ dna, const u8 *__restrict__ qs, const u64 *__restrict__ roff, u64 N,
                                                     u8 *__restrict__ bases, u8 *__restrict__ quals)
{
    const u32 sub = threadIdx.x & 15u;
    const u64 ngrp = ((u64)gridDim.x * blockDim.x) >> 4;
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < N; i += ngrp) {
        const u64 b = roff[i], len = roff[i + 1] - b, o = b + i;
        for (u64 k = sub; k < len; k += 16) { bases[b + k] = dna[o + k]; quals[b + k] = qs[o + k]; }
    }
}
static void steps34_positions(bfq_ctx *c, const u8 *d_bases, const u8 *d_quals, const u64 *d_roff, u8 *d_out_bases, u8 *d_out_quals)
{
    const u64 n = c->n, N = c->N;
    if (!n) return;
    u8 *dna = c->alloc<u8>(n + 64), *qs = c->alloc<u8>(n + 64);
    if (N) KLAUNCH(c, K_MISC, 4.0 * (double)(n - N), k_lines_init, bfq_grid(N, 16), 256, d_bases, d_quals, d_roff, N, c->P.B, dna, qs);
    u8 *in = c->alloc<u8>(n + 64);
    bfq_lcp_flags(c, c->d_lcp, n, c->P.K, in);
    ClusterPos pm{c->d_w12, c->d_text3, dna, qs, c->P.B};
    RankIndex none{nullptr, n};
    bfq_clusters(c, none, c->d_bwt, c->d_qual, in, n, &pm);
    if (N) KLAUNCH(c, K_MISC, 4.0 * (double)(n - N), k_lines_strip, bfq_grid(N, 16), 256, (const u8 *)dna, (const u8 *)qs, d_roff, N, d_out_bases, d_out_quals);
}

// ---- the whole path under a workspace cap (the counterpart of bfq_ext's pile files, bfq_ext.cpp:190-348, and
// of its lock-step inversion, decode.cpp:499-966): what bounds a block is the eBWT-sized state -- 24 B of sort records, the
// 8-byte LF entry, the eBWT / QS / LCP arrays.  None of it is needed all at once.  The suffixes are cut into piles by their
// first TWO symbols (clusters never cross such a pile when K >= 2: the LCP at its border is at most 1); a pile is sorted,
// refined and analysed on its own, every row still knows the text position of its suffix (the sort payload), so edits go
// straight to that position of the output line streams and the one place that needs bwt[LF(row)] (bfq_int.cpp:545-560)
// reads the packed text two symbols back.  No eBWT is ever materialised, nothing is inverted: the line streams ARE the
// reads.  Resident: 4.4 n bytes + one pile.  dna / qs: n bytes each (read i at roff[i] + i, then '\n').
static void steps_capped(bfq_ctx *c, const u8 *d_bases, const u8 *d_quals, const u64 *d_roff, u64 N, u64 total, u8 *dna, u8 *qs)
{
    const u64 n = total + N;
    if (n >= (1ull << BFQ_POS_BITS)) throw BfqError{BFQ_E_ARG, "collection too large (2^37 rows)"};
    c->n = n; c->N = N;
    c->d_bwt = c->d_qual = nullptr; c->d_lcp = nullptr; c->d_gcnt = nullptr;     // no eBWT exists in this mode
    if (!n) return;
    if (c->P.K < 1) throw BfqError{BFQ_E_NOMEM, "the capped mode needs -k >= 1 (with -k 0 every row is in one cluster): raise the workspace cap"};
    const size_t m0 = c->mark();
    const u64 nwords = n / BFQ_SYMS_PER_WORD + 3;
    u64 *text3 = c->alloc<u64>(bfq_t3_alloc(nwords));
    u8 *T8 = c->alloc<u8>(n + 64), *Q8 = c->alloc<u8>(n + 64);
    bfq_build_text(c, d_bases, d_quals, d_roff, N, n, T8, Q8, text3, nwords);
    if (N) KLAUNCH(c, K_MISC, 4.0 * (double)(n - N), k_lines_init, bfq_grid(N, 16), 256, d_bases, d_quals, d_roff, N, c->P.B, dna, qs);
    u64 cnt[36];
    bfq_pile_pair_counts(c, T8, n, cnt);                         // synchronises
    c->fetchCounters();
    if (c->h_cnt.errSymbol || c->h_cnt.errTooLong) { c->release(m0); return; }   // reported by the caller's check
    const bool two = c->P.K >= 2;                                // -k 1: clusters may cross two-symbol piles: first-symbol piles
    for (u32 s = 1; s <= 5; s++) {
        for (u32 s2 = two ? 1u : 7u; s2 <= (two ? 5u : 7u); s2++) {
            u64 rows = 0;
            if (two) rows = cnt[6 * s + s2]; else for (int q = 0; q < 6; q++) rows += cnt[6 * s + q];
            if (!rows) continue;
            if (rows > c->cappedPileRows) {
                char b[220];
                snprintf(b, sizeof b, "pile '%c%c' holds %llu rows, the workspace cap leaves room for %llu: raise bfq_params.ws_cap_mib / BFQ_WS_CAP",
                         "#ACGNT"[s], two ? "#ACGNT"[s2] : '*', (unsigned long long)rows, (unsigned long long)c->cappedPileRows);
                throw BfqError{BFQ_E_NOMEM, b};
            }
            const size_t mp = c->mark();
            PileRows pr;
            if (bfq_env().trace) fprintf(stderr, "[bfq capped] pile %c%c: %llu rows, arena %zu of %zu used\n", "#ACGNT"[s], two ? "#ACGNT"[s2] : '*', (unsigned long long)rows, c->wsTop, c->wsCap);
            const u64 m = bfq_run_one_pile(c, T8, Q8, text3, n, s, s2, c->P.term & 0xFF, &pr);
            if (bfq_env().trace) { c->sync(); fprintf(stderr, "[bfq capped]   sorted + refined\n"); }
            if (m) {
                u8 *in = c->alloc<u8>(m + 64);
                bfq_lcp_flags(c, pr.lcp, m, c->P.K, in);
                ClusterPos pm{pr.w12, text3, dna, qs, c->P.B};
                RankIndex none{nullptr, m};
                bfq_clusters(c, none, pr.bwt, pr.qs, in, m, &pm);
            }
            c->sync();                                           // the pile's arrays go back to the arena
            if (bfq_env().trace) fprintf(stderr, "[bfq capped]   clusters done\n");
            c->release(mp);
        }
    }
    c->release(m0);
}


extern "C" int bfq_run_reads_device(bfq_ctx *c, const uint8_t *d_bases, const uint8_t *d_quals,
                                    const uint64_t *d_read_off, uint64_t N, uint64_t total, uint8_t *d_out_bases,
                                    uint8_t *d_out_quals, bfq_stats *st)
{
    return guarded(c, [&] {
        if (st) memset(st, 0, sizeof *st);
        reserve_step1(c, total + N, N, 0);
        c->zeroCounters();
        if (c->capped) {
            const u64 n = total + N;
            u8 *dna = c->alloc<u8>(n + 64), *qs = c->alloc<u8>(n + 64);
            steps_capped(c, d_bases, d_quals, (const u64 *)d_read_off, N, total, dna, qs);
            if (N) KLAUNCH(c, K_MISC, 4.0 * (double)total, k_lines_strip, bfq_grid(N, 16), 256, (const u8 *)dna, (const u8 *)qs, (const u64 *)d_read_off, N, d_out_bases, d_out_quals);
            c->fetchCounters();
            c->profCollect();
            check_counters(c);
            fill_stats(c, st);
            return;
        }
        const bool posMode = !c->piles && c->env.posMode;
        c->keepRecs = posMode;
        bfq_step1_device(c, d_bases, d_quals, (const u64 *)d_read_off, N, total, c->P.term, st);
        c->keepRecs = false;
        if (posMode) steps34_positions(c, d_bases, d_quals, (const u64 *)d_read_off, d_out_bases, d_out_quals);
        else steps234_device(c, (u64 *)d_read_off, d_out_bases, d_out_quals);
        c->fetchCounters();
        c->profCollect();
        check_counters(c);
        fill_stats(c, st);
    });
}

extern "C" int bfq_run_reads(bfq_ctx *c, const uint8_t *h_bases, const uint8_t *h_quals, const uint64_t *h_read_off,
                             uint64_t N, uint8_t *h_out_bases, uint8_t *h_out_quals, bfq_stats *st)
{
    return guarded(c, [&] {
        if (st) memset(st, 0, sizeof *st);
        if (!h_read_off) throw BfqError{BFQ_E_ARG, "null read offsets"};
        u64 total = h_read_off[N];
        reserve_step1(c, total + N, N, 6 * (total + 256) + 10 * (N + 64));   // (+ 2 n: the capped mode's line streams beside its outputs)
        c->zeroCounters();
        u8 *db = c->alloc<u8>(total + 64), *dq = c->alloc<u8>(total + 64);
        u8 *ob = c->alloc<u8>(total + 64), *oq = c->alloc<u8>(total + 64);
        u64 *dr = c->alloc<u64>(N + 1);
        bfq_upload(c, db, h_bases, total);
        bfq_upload(c, dq, h_quals, total);
        bfq_upload(c, dr, h_read_off, 8 * (N + 1));
        if (c->capped) {
            u8 *dna = c->alloc<u8>(total + N + 64), *qs = c->alloc<u8>(total + N + 64);
            steps_capped(c, db, dq, dr, N, total, dna, qs);
            if (N) KLAUNCH(c, K_MISC, 4.0 * (double)total, k_lines_strip, bfq_grid(N, 16), 256, (const u8 *)dna, (const u8 *)qs, (const u64 *)dr, N, ob, oq);
        } else {
            bfq_step1_device(c, db, dq, dr, N, total, c->P.term, st);
            steps234_device(c, dr, ob, oq);
        }
        bfq_download(c, h_out_bases, ob, total);
        bfq_download(c, h_out_quals, oq, total);
        c->fetchCounters();
        c->profCollect();
        check_counters(c);
        fill_stats(c, st);
    });
}

extern "C" int bfq_build_ebwt(bfq_ctx *c, const uint8_t *h_bases, const uint8_t *h_quals, const uint64_t *h_read_off,
                              uint64_t N, int term_out, uint8_t *h_bwt, uint8_t *h_bwtqs, uint16_t *h_lcp16)
{
    return guarded(c, [&] {
        if (!h_read_off) throw BfqError{BFQ_E_ARG, "null read offsets"};
        u64 total = h_read_off[N], n = total + N;
        reserve_step1(c, n, N, 2 * (total + 256) + 8 * (N + 64), false);
        c->zeroCounters();
        u8 *db = c->alloc<u8>(total + 64), *dq = c->alloc<u8>(total + 64);
        u64 *dr = c->alloc<u64>(N + 1);
        bfq_upload(c, db, h_bases, total);
        bfq_upload(c, dq, h_quals, total);
        bfq_upload(c, dr, h_read_off, 8 * (N + 1));
        bfq_step1_device(c, db, dq, dr, N, total, term_out, nullptr);
        if (h_bwt) bfq_download(c, h_bwt, c->d_bwt, n);
        if (h_bwtqs) bfq_download(c, h_bwtqs, c->d_qual, n);
        if (h_lcp16) bfq_download(c, h_lcp16, c->d_lcp, 2 * n);
        c->fetchCounters();
        c->profCollect();
        check_counters(c);
    });
}

extern "C" int bfq_fetch_ebwt(bfq_ctx *c, uint8_t *h_bwt, uint8_t *h_qs, uint16_t *h_lcp16)
{
    return guarded(c, [&] {
        if (!c->d_bwt) throw BfqError{BFQ_E_ARG, "no eBWT resident"};
        if (h_bwt) HIP_CHECK(hipMemcpyAsync(h_bwt, c->d_bwt, c->n, hipMemcpyDeviceToHost, c->stream));
        if (h_qs) HIP_CHECK(hipMemcpyAsync(h_qs, c->d_qual, c->n, hipMemcpyDeviceToHost, c->stream));
        if (h_lcp16) HIP_CHECK(hipMemcpyAsync(h_lcp16, c->d_lcp, 2 * c->n, hipMemcpyDeviceToHost, c->stream));
        c->sync();
    });
}

__global__ __launch_bounds__(256) void k_lcp_widen(const u8 *__restrict__ raw, int lb, u64 n, u16 *__restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 v = lb == 1 ? raw[i] : lb == 2 ? ((const u16 *)raw)[i] : ((const u32 *)raw)[i];
        out[i] = (u16)(v > 0xFFFE ? 0xFFFE : v);
    }
}

__global__ __launch_bounds__(256) void k_count_byte(const u8 *__restrict__ a, u64 n, u32 v, u64 *out)
{
    u64 k = 0;
    const u64 n16 = n / 16;                                       // the buffer is 16-byte aligned
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) {
        const uint4 x = ((const uint4 *)a)[i];
        const u32 w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int b = 0; b < 4; b++) k += (((w[q] >> (8 * b)) & 0xFFu) == v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 15)) k += (a[n16 * 16 + threadIdx.x] == (u8)v);
    k = bfq_readlane64(bfq_wave_incscan64(k), 63);
    if ((threadIdx.x & 63) == 0 && k) atomicAdd((unsigned long long *)out, (unsigned long long)k);
}

// steps 2-4 on an eBWT given by the caller (host memory or files); leaves the smoothed reads on the device (arena).
// The eBWT and its qualities are uploaded into the context's text buffer (outside the arena), the terminators are
// counted there, and only then is the arena sized.
struct SmoothOut { u8 *ob, *oq; u64 *roff; u64 N, total; };
// hostOut != nullptr: the reads go to these host arrays while the inversion is still running (read-range chunks, each copied
// out on the copy stream behind its walk); the caller then downloads nothing but the offsets
static void smooth_invert_core(bfq_ctx *c, HostRef h_bwt, HostRef h_bwtqs, HostRef h_lcp, int lcp_bytes,
                               uint64_t n, size_t extraWs, bfq_stats *st, SmoothOut *res, const StreamOut *hostOut = nullptr)
{
    {
        if (st) memset(st, 0, sizeof *st);
        if (n && (h_bwt.null() || h_bwtqs.null())) throw BfqError{BFQ_E_ARG, "null eBWT"};
        const bool haveLcp = !h_lcp.null();
        if (haveLcp && lcp_bytes != 1 && lcp_bytes != 2 && lcp_bytes != 4) throw BfqError{BFQ_E_ARG, "lcp_bytes must be 1, 2 or 4"};
        const u64 npad = (n + 255) & ~255ull;
        bfq_phase("alloc");
        // the arena of the table path, allocated beside the upload of the eBWT (its exact size needs the number of reads,
        // which the upload is counted for: reads of 32 bases or more assumed, reserve() below allocates again when that was short)
        if (!c->env.compact && n >= (64u << 20) && !c->env.noOverlap)
            c->reserveBegin(ws_need_given(n, n / 32, extraWs + (haveLcp ? (size_t)lcp_bytes * n : 0)));
        struct PendGuard { bfq_ctx *c; ~PendGuard() { c->reserveJoin(false, 0); } } pendGuard{c};
        u8 *in_bwt = c->textBuf(2 * npad + 256), *in_qs = in_bwt + npad;
        bfq_phase("read_h2d");
        bfq_upload(c, in_bwt, h_bwt, n);
        c->zeroCounters();
        if (n) KLAUNCH(c, K_MISC, (double)n, k_count_byte, (unsigned)(n / 4096 + 1 < 2048 ? n / 4096 + 1 : 2048), 256, (const u8 *)in_bwt, n, (u32)(c->P.term & 0xFF), &c->d_cnt->pad[0]);
        c->fetchCounters();
        const u64 N = c->h_cnt.pad[0];
        if (n && N == 0) throw BfqError{BFQ_E_NOT_EBWT, "no terminator in the eBWT"};
        u64 total = n - N;
        bfq_phase("alloc");
        bool compact = c->env.compact;
        if (!compact) {
            try { c->reserve(ws_need_given(n, N, extraWs + (haveLcp ? (size_t)lcp_bytes * n : 0))); }
            catch (const BfqError &e) { if (e.code != BFQ_E_NOMEM || !c->wsLimit()) throw; compact = true; }
        }
        if (compact) {
            // under a workspace cap: no LF table -- rank blocks, qualities edited in place, a replacement array; the LCP file
            // streamed through a window, or the LCP deduced with a ring queue and dropped once the flags exist (k_compact.hip):
            // 5 (bfq_ext) / 8 (bfq_int) bytes per row + the outputs instead of 17
            // the cap counts the eBWT and the qualities too (the text buffer, outside the arena)
            const size_t need = bfq_ws_need_compact(c, n, N, extraWs, haveLcp, c->textCap);
            if (c->wsLimit() && need + c->textCap > c->wsLimit()) {
                char b[200];
                snprintf(b, sizeof b, "device memory of %.1f GiB (eBWT + qualities + workspace) is above the cap of %.1f GiB (bfq_params.ws_cap_mib / BFQ_WS_CAP)",
                         (need + c->textCap) / 1073741824.0, c->wsLimit() / 1073741824.0);
                throw BfqError{BFQ_E_NOMEM, b};
            }
            c->reserve(need);
            bfq_phase("read_h2d");
            bfq_upload(c, in_qs, h_bwtqs, n);
            u8 *ob = nullptr, *oq = nullptr;
            u64 *d_roff = c->alloc<u64>(N + 1);
            u32 *lens = c->alloc<u32>(N + 1);
            c->d_bwt = in_bwt; c->d_qual = in_qs; c->d_lcp = nullptr; c->d_gcnt = nullptr; c->gcntTerm = -1;
            bfq_phase("gpu");
            bfq_steps234_compact(c, in_bwt, in_qs, h_lcp, lcp_bytes, n, N, d_roff, lens, &ob, &oq, extraWs, c->textCap);
            if (hostOut) {
                if (hostOut->h_dna) bfq_download(c, hostOut->h_dna, ob, total);
                if (hostOut->h_qs) bfq_download(c, hostOut->h_qs, oq, total);
            }
            res->ob = ob; res->oq = oq; res->roff = d_roff; res->N = N; res->total = total;
            return;
        }
        bfq_phase("read_h2d");
        // the qualities are not needed before the LF table is built: a pageable / file source is staged by a helper thread
        // while this thread goes on (the LCP deduction from the BWT alone takes longer than the upload)
        BfqAsyncUpload *qsUp = nullptr;
        const bool qsPinned = h_bwtqs.ptr && bfq_is_pinned(h_bwtqs.ptr);
        const bool qsAsync = !haveLcp && n >= (64u << 20) && !c->env.noOverlap;
        hipEvent_t qsEv = nullptr;
        if (qsAsync && qsPinned) {                                 // one DMA on the copy stream, beside the kernels of the LCP deduction
            if (!c->copyStream) HIP_CHECK(hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&qsEv, hipEventDisableTiming));
            HIP_CHECK(hipMemcpyAsync(in_qs, h_bwtqs.ptr, n, hipMemcpyHostToDevice, c->copyStream));
            HIP_CHECK(hipEventRecord(qsEv, c->copyStream));
        } else if (qsAsync) qsUp = bfq_upload_begin(c, in_qs, h_bwtqs, n);
        else bfq_upload(c, in_qs, h_bwtqs, n);
        struct EvGuard { hipEvent_t &e; bfq_ctx *c; ~EvGuard() { if (e) { (void)hipStreamSynchronize(c->copyStream); (void)hipEventDestroy(e); e = nullptr; } } } evGuard{qsEv, c};
        struct Join { BfqAsyncUpload *u; ~Join() { if (u) { try { bfq_upload_join(u); } catch (...) {} } } } joinGuard{qsUp};
        u8 *ob = c->alloc<u8>(total + 64), *oq = c->alloc<u8>(total + 64);
        u64 *d_roff = c->alloc<u64>(N + 1);
        u32 *lens = c->alloc<u32>(N + 1);
        // the persistent arrays hold the given eBWT directly
        c->n = n; c->N = N;
        c->d_bwt = in_bwt; c->d_qual = in_qs;
        c->d_lcp = c->alloc<u16>(n + 64);
        c->d_gcnt = c->alloc<u32>(6 * (n / 256 + 1));
        c->gcntTerm = -1;
        if (haveLcp) {
            // explicit LCP (bfq_ext): the file's entries (1, 2 or 4 bytes) are widened / clamped to 16 bits on the device
            size_t mr = c->mark();
            u8 *raw = c->alloc<u8>((size_t)lcp_bytes * n + 64);
            bfq_upload(c, raw, h_lcp, (size_t)lcp_bytes * n);
            KLAUNCH(c, K_MISC, (double)(lcp_bytes + 2) * (double)n, k_lcp_widen, bfq_grid(n, 256), 256, (const u8 *)raw, lcp_bytes, n, c->d_lcp);
            c->release(mr);
        } else {
            // bfq_int deduces the LCP from the BWT alone (detect_minima, bfq_int.cpp:183-300): interval refinement, k_bfs.hip
            bfq_phase("gpu");
            bfq_lcp_from_bwt(c, in_bwt, n, N, c->P.term & 0xFF, c->d_lcp, c->d_gcnt);
            c->gcntTerm = c->P.term & 0xFF;
        }
        if (qsUp) { bfq_phase("read_h2d"); joinGuard.u = nullptr; bfq_upload_join(qsUp); }
        if (qsEv) HIP_CHECK(hipStreamWaitEvent(c->stream, qsEv, 0));
        bfq_phase("gpu");
        steps234_device(c, d_roff, ob, oq, lens, hostOut);
        res->ob = ob; res->oq = oq; res->roff = d_roff; res->N = N; res->total = total;
    }
}

extern "C" int bfq_smooth_invert(bfq_ctx *c, const uint8_t *h_bwt, const uint8_t *h_bwtqs, const void *h_lcp,
                                 int lcp_bytes, uint64_t n, uint8_t *h_out_bases, uint8_t *h_out_quals,
                                 uint64_t *h_out_read_off, bfq_stats *st)
{
    return guarded(c, [&] {
        SmoothOut r;
        const StreamOut ho{h_out_bases, h_out_quals, true};
        smooth_invert_core(c, HostRef::mem(h_bwt), HostRef::mem(h_bwtqs), HostRef::mem(h_lcp), lcp_bytes, n, 0, st, &r, &ho);
        bfq_download(c, h_out_read_off, r.roff, 8 * (r.N + 1));
        c->fetchCounters();
        c->profCollect();
        check_counters(c);
        fill_stats(c, st);
    });
}

// ---------------------------------------------------------------- FASTQ text in / out
// The FASTQ text lives outside the arena (its record count sizes the arena) in a buffer the context keeps
// between calls: uploaded first, lines counted on the device, then the workspace is reserved for the real N.
u64 bfq_fastq_count_lines(bfq_ctx *c, const u8 *d_buf, u64 len);   // k_fastq.hip
u8 *bfq_ctx::textBuf(size_t bytes)
{
    residentValid = false;                                      // whoever asks is about to overwrite the buffer
    if (bytes > textCap) {
        if (d_text) { HIP_CHECK(hipStreamSynchronize(stream)); HIP_CHECK(hipFree(d_text)); d_text = nullptr; textCap = 0; }
        size_t want = (bytes + (bytes >> 4) + 0xFFFFF) & ~(size_t)0xFFFFF;
        if (hipMalloc((void **)&d_text, want) != hipSuccess) {
            (void)hipGetLastError();
            d_text = nullptr;
            throw BfqError{BFQ_E_NOMEM, "device buffer for the FASTQ text"};
        }
        textCap = want;
    }
    return d_text;
}

// Uploads the parts back to back (a part that does not end in '\n' gets one, so that no record straddles two
// parts); pstart[p] = offset of part p in the device text, pstart[nparts] = its length.
struct TextSrc { HostRef ref; u64 len; };
static bool src_ends_with_newline(const TextSrc &t)
{
    if (!t.len) return true;
    if (t.ref.ptr) return ((const u8 *)t.ref.ptr)[t.len - 1] == (u8)'\n';
    u8 b = 0;
    if (pread(t.ref.fd, &b, 1, (off_t)(t.ref.off + t.len - 1)) != 1) throw BfqError{BFQ_E_IO, "cannot read the input file"};
    return b == (u8)'\n';
}
static u8 *fastq_upload_and_reserve(bfq_ctx *c, const TextSrc *parts, int nparts, std::vector<u64> &pstart, size_t extraWs = 0, bool allowCapped = true)
{
    pstart.assign(nparts + 1, 0);
    std::vector<u8> addNl(nparts, 0);
    u64 len = 0;
    for (int p = 0; p < nparts; p++) {
        if (parts[p].len && parts[p].ref.null()) throw BfqError{BFQ_E_ARG, "null FASTQ text"};
        pstart[p] = len;
        len += parts[p].len;
        if (!src_ends_with_newline(parts[p])) { addNl[p] = 1; len++; }
    }
    pstart[nparts] = len;
    bfq_phase("alloc");
    u8 *d_fq = c->textBuf(len + 64);
    bfq_phase("read_h2d");
    for (int p = 0; p < nparts; p++) {
        bfq_upload(c, d_fq + pstart[p], parts[p].ref, parts[p].len);
        if (addNl[p]) HIP_CHECK(hipMemsetAsync(d_fq + pstart[p] + parts[p].len, '\n', 1, c->stream));
    }
    bfq_phase("alloc");
    c->reserve(16 * (len / 4096 + 16) + (64u << 20));
    bfq_phase("gpu");
    u64 nlines = bfq_fastq_count_lines(c, d_fq, len);
    u64 N = nlines / 4 + 1;
    u64 nb = len / 2 + 1;                                       // rows <= bytes / 2
    bfq_phase("alloc");
    reserve_step1(c, nb, N, 3 * (len + 4096) + 128 * (N + 64) + 8 * (nlines + 64) + extraWs, allowCapped);
    bfq_phase("gpu");
    return d_fq;
}

extern "C" uint64_t bfq_fastq_out_bound(uint64_t total_bases, uint64_t n_reads, uint64_t header_bytes)
{
    return 2 * total_bases + 5 * n_reads + (header_bytes ? header_bytes : n_reads);
}

__global__ __launch_bounds__(256) void k_lcp_narrow(const u16 *__restrict__ lcp, int lb, u64 n, u8 *__restrict__ raw)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 v = lcp[i];
        if (lb == 1) raw[i] = (u8)(v > 255u ? 255u : v);         // eGap --lbytes 1 saturates
        else ((u32 *)raw)[i] = v;
    }
}

static void fastq_build_ebwt_core(bfq_ctx *c, TextSrc text, int term_out, HostRef bwt, HostRef qs, HostRef lcp, int lcp_bytes,
                                  uint64_t cap_rows, uint64_t *n_rows, uint64_t *n_reads)
{
    if (!lcp.null() && lcp_bytes != 1 && lcp_bytes != 2 && lcp_bytes != 4) throw BfqError{BFQ_E_ARG, "lcp_bytes must be 1, 2 or 4"};
    std::vector<u64> ps;
    u8 *d_fq = fastq_upload_and_reserve(c, &text, 1, ps, 0, false);
    c->zeroCounters();
    DevFastq fq;
    bfq_fastq_parse(c, d_fq, ps[1], &fq);
    u64 n = fq.total + fq.N;
    if (n_rows) *n_rows = n;
    if (n_reads) *n_reads = fq.N;
    if (n > cap_rows) throw BfqError{BFQ_E_ARG, "output buffers smaller than the eBWT (need total bases + reads entries)"};
    bfq_step1_device(c, fq.bases, fq.quals, fq.roff, fq.N, fq.total, term_out, nullptr);
    if (!bwt.null()) bfq_download(c, bwt, c->d_bwt, n);
    if (!qs.null()) bfq_download(c, qs, c->d_qual, n);
    if (!lcp.null()) {
        if (lcp_bytes == 2) bfq_download(c, lcp, c->d_lcp, 2 * n);
        else {
            u8 *raw = c->alloc<u8>((size_t)lcp_bytes * n + 64);
            if (n) KLAUNCH(c, K_MISC, (double)(lcp_bytes + 2) * (double)n, k_lcp_narrow, bfq_grid(n, 256), 256, (const u16 *)c->d_lcp, lcp_bytes, n, raw);
            bfq_download(c, lcp, raw, (size_t)lcp_bytes * n);
        }
    }
    c->fetchCounters();
    c->profCollect();
    check_counters(c);
}

// ---- the one-shot tools (gsufsort / eGap / bfq_int / bfq_ext processes): files in, files out.
// An output file: mapped and pre-faulted in the background when it is a regular file, written with pwrite otherwise.
struct OutFile {
    int fd = -1;
    bfq_outmap *m = nullptr;
    HostRef at(u64 off) const
    {
        if (!m) return HostRef::file(fd, off);
        HostRef h = HostRef::mem(bfq_outmap_ptr(m) + off);
        h.om = m; h.off = off;
        return h;
    }
    void open(int f, u64 mapLen, u64 prefault)
    {
        fd = f;
        if (f < 0) return;
        m = bfq_outmap_take(f, mapLen);
        if (m) bfq_outmap_extend(m, prefault);
        else m = bfq_outmap_open(f, mapLen, prefault);
    }
    bool close(u64 finalLen)
    {
        bool ok = true;
        if (m) ok = bfq_outmap_close(m, finalLen);
        else if (fd >= 0) ok = ftruncate(fd, (off_t)finalLen) == 0 || errno == EINVAL;   // EINVAL: not a regular file
        m = nullptr;
        return ok;
    }
};
static void fastq_build_ebwt_oneshot(bfq_ctx *c, int fastq_fd, uint64_t len, int term_out, int bwt_fd, int qs_fd, int lcp_fd, int lcp_bytes,
                                     uint64_t *n_rows, uint64_t *n_reads)
{
    const bool wantLcp = lcp_fd >= 0;
    if (wantLcp && lcp_bytes != 1 && lcp_bytes != 2 && lcp_bytes != 4) throw BfqError{BFQ_E_ARG, "lcp_bytes must be 1, 2 or 4"};
    TextSrc text{HostRef::file(fastq_fd), len};
    // The outputs are sized to their bound (rows <= bytes / 2) and pre-faulted in the background -- from the moment the tool
    // opened them (bfq_output_prefault) or from here: by the time the first pile is sorted the page-cache pages exist and the
    // copy out of the staging buffers runs at memcpy speed.
    const u64 capRows = len / 2 + 64, est = bfq_fastq_rows_estimate(fastq_fd, len);
    OutFile ob, oq, ol;
    ob.open(bwt_fd, capRows, est);
    oq.open(qs_fd, capRows, est);
    if (wantLcp) ol.open(lcp_fd, capRows * (u64)lcp_bytes, est * (u64)lcp_bytes);
    u64 n = 0;
    bool done = false;
    // HBM of this process, and why it is cut into pieces: the driver scrubs freed HBM at ~30 GB/s and the NEXT process's
    // allocations wait for it (profiles/microbench/alloc_after_exit.hip; profiles/r3/dropin_phases.md: bfq_int waited 1.3-4.3 s
    // behind a gsufsort that had held 102 GiB).  So every piece is as small as its contents and goes back the moment it is dead:
    //   text buffer   the FASTQ text, then (the reads gathered) the eBWT and its qualities until they are written
    //   parse arena   line index, records, gathered reads: dead once the terminated text is built -- and then
    //   arena         sort records + histograms of ONE pile, sized from the actual pile sizes, in the parse arena's memory
    //                 (what the next process waits for is everything this one ever gave back, not what it held last)
    //   text arrays   T8 / Q8 / packed text: until the last pile is sorted
    char *pws = nullptr, *tws = nullptr;
    char *ws0 = c->ws; size_t cap0 = c->wsCap;
    bool swapped = false;
    auto restoreArena = [&] { if (swapped) { c->ws = ws0; c->wsCap = cap0; c->wsTop = 0; swapped = false; } };
    auto finish = [&](bool ok) {
        c->onRows = nullptr; c->extBwt = c->extQual = nullptr; c->lcpScratch = false;
        if (!ok) { try { bfq_write_wait(c); } catch (...) {} }
        (void)hipStreamSynchronize(c->stream);
        restoreArena();
        if (pws) { (void)hipFree(pws); pws = nullptr; }
        if (tws) { (void)hipFree(tws); tws = nullptr; }
        const bool a = ob.close(ok ? n : 0), b = oq.close(ok ? n : 0), l = ol.close(ok ? n * (u64)lcp_bytes : 0);
        if (ok && !(a && b && l)) throw BfqError{BFQ_E_IO, "cannot size the output files"};
    };
    try {
        bfq_phase("alloc");
        const bool addNl = !src_ends_with_newline(text);
        const u64 tl = len + (addNl ? 1 : 0);
        u8 *d_fq = c->textBuf(tl + 64);
        bfq_phase("read_h2d");
        bfq_upload(c, d_fq, text.ref, len);
        if (addNl) HIP_CHECK(hipMemsetAsync(d_fq + len, '\n', 1, c->stream));
        bfq_phase("alloc");
        c->reserve(16 * (tl / 4096 + 16) + (64u << 20));
        bfq_phase("gpu");
        const u64 nlines = bfq_fastq_count_lines(c, d_fq, tl);
        const u64 Nb = nlines / 4 + 1;
        bfq_phase("alloc");
        const size_t parseBytes = (tl + 8192) + 128 * (Nb + 64) + 8 * (nlines + 64) + 16 * (tl / 4096 + 16) + (64u << 20);
        if (hipMalloc((void **)&pws, parseBytes) != hipSuccess) { (void)hipGetLastError(); pws = nullptr; throw BfqError{BFQ_E_NOMEM, "device buffer for the parsed records"}; }
        ws0 = c->ws; cap0 = c->wsCap;                                          // (the small arena of the line count)
        c->ws = pws; c->wsCap = parseBytes; c->wsTop = 0; swapped = true;      // ws0 comes back below
        bfq_phase("gpu");
        c->zeroCounters();
        DevFastq fq;
        bfq_fastq_parse(c, d_fq, tl, &fq);
        n = fq.total + fq.N;
        const u64 N = fq.N;
        if (n_rows) *n_rows = n;
        if (n_reads) *n_reads = N;
        c->writeHint = (size_t)n * (2 + (wantLcp ? (size_t)lcp_bytes : 0));
        const u64 npad = (n + 64 + 255) & ~255ull;
        const bool ext = 2 * npad <= c->textCap;                // always (rows <= bytes / 2); the FASTQ text is dead from here on
        u8 *raw = nullptr;
        auto hook = [&](u64 start, u64 m) {                     // rows [start, start + m) are final: out they go while the next pile is sorted
            if (!m) return;
            if (ob.fd >= 0) bfq_write_async(c, ob.at(start), c->d_bwt + start, m);
            if (oq.fd >= 0) bfq_write_async(c, oq.at(start), c->d_qual + start, m);
            if (wantLcp) {
                if (lcp_bytes == 2) bfq_write_async(c, ol.at(2 * start), c->d_lcp + start, 2 * m);
                else {
                    u8 *r = raw + (size_t)lcp_bytes * start;
                    KLAUNCH(c, K_MISC, (double)(lcp_bytes + 2) * (double)m, k_lcp_narrow, bfq_grid(m, 256), 256, (const u16 *)(c->d_lcp + start), lcp_bytes, m, r);
                    bfq_write_async(c, ol.at((u64)lcp_bytes * start), r, (size_t)lcp_bytes * m);
                }
            }
        };
        const size_t lcpExtra = wantLcp ? (size_t)(lcp_bytes == 2 ? 0 : lcp_bytes) * (n + 256) + 4096 : 0;
        if (n < (32u << 20)) {
            // a small collection: one piece in one arena (0.9 GB at most); the parsed reads stay where they are meanwhile
            restoreArena();
            bfq_phase("alloc");
            c->reserve(ws_need(n, N, lcpExtra));
            bfq_phase("gpu");
            c->piles = false;
            if (ext) { c->extBwt = c->d_text; c->extQual = c->d_text + npad; }
            if (wantLcp && lcp_bytes != 2) raw = c->alloc<u8>((size_t)lcp_bytes * n + 64);
            c->onRows = hook;
            bfq_step1_device(c, fq.bases, fq.quals, fq.roff, N, fq.total, term_out, nullptr);
        } else {
            const u64 nwords = n / BFQ_SYMS_PER_WORD + 3;
            const size_t w3 = (8 * bfq_t3_alloc(nwords) + 255) & ~(size_t)255;
            bfq_phase("alloc");
            if (hipMalloc((void **)&tws, 2 * npad + w3 + 8 * (N + 2) + 4096) != hipSuccess) { (void)hipGetLastError(); tws = nullptr; throw BfqError{BFQ_E_NOMEM, "device buffer for the text arrays"}; }
            bfq_phase("gpu");
            PileText pt{(u8 *)tws, (u8 *)tws + npad, (u64 *)(tws + 2 * npad)};
            u64 *roff2 = (u64 *)(tws + 2 * npad + w3);
            bfq_build_text(c, fq.bases, fq.quals, fq.roff, N, n, pt.T8, pt.Q8, pt.text3, nwords);
            HIP_CHECK(hipMemcpyAsync(roff2, fq.roff, 8 * (N + 1), hipMemcpyDeviceToDevice, c->stream));
            u64 cnt[36];
            bfq_pile_pair_counts(c, pt.T8, n, cnt);             // synchronises
            c->fetchCounters();
            check_counters(c);                                  // forbidden symbols, reads beyond BFQ_MAX_READ_LEN
            restoreArena();                                     // the parsed reads are dead
            // piles above an eighth of the rows are split by their second symbol; the arena holds the largest piece
            const u64 capTarget = n / 8 + (1u << 20);
            u64 cap = 1u << 20;
            for (int s1 = 1; s1 <= 5; s1++) {
                u64 tot = 0, big = 0;
                for (int s2 = 0; s2 <= 5; s2++) { tot += cnt[6 * s1 + s2]; big = std::max(big, cnt[6 * s1 + s2]); }
                cap = std::max(cap, tot <= capTarget ? tot : big);
            }
            bfq_phase("alloc");
            const size_t needPiles = bfq_ws_need_piles(n, N, cap, lcpExtra + (wantLcp ? 2 * (n + 256) : 0), true);
            if (parseBytes >= needPiles + (1u << 20) && !(c->wsLimit() && parseBytes > c->wsLimit())) {
                c->wsFree();                                    // the pile arena lives where the parsed reads were
                c->ws = pws; c->wsCap = parseBytes; c->wsTop = 0; pws = nullptr;
            } else { (void)hipFree(pws); pws = nullptr; }
            c->reserve(needPiles);
            bfq_phase("gpu");
            c->piles = true;
            c->lcpScratch = !wantLcp;
            if (ext) { c->extBwt = c->d_text; c->extQual = c->d_text + npad; }
            else { c->extBwt = nullptr; c->extQual = nullptr; }
            if (wantLcp && lcp_bytes != 2) raw = c->alloc<u8>((size_t)lcp_bytes * n + 64);
            c->onRows = hook;
            bfq_step1_piles(c, nullptr, nullptr, roff2, N, fq.total, term_out, nullptr, &pt, capTarget);
        }
        c->onRows = nullptr;
        c->fetchCounters();                                     // waits for the stream
        c->profCollect();
        check_counters(c);
        bfq_phase("d2h_write");
        if (ext && !wantLcp) {                                  // what is still being written lives in the text buffer: the rest goes back now
            c->dropWorkspace();
            if (tws) { (void)hipFree(tws); tws = nullptr; }
            if (pws) { (void)hipFree(pws); pws = nullptr; }
        }
        bfq_write_wait(c);
        done = true;
    } catch (...) {
        finish(false);
        throw;
    }
    if (done) finish(true);
}

extern "C" int bfq_fastq_build_ebwt(bfq_ctx *c, const uint8_t *h_fastq, uint64_t len, int term_out, uint8_t *h_bwt,
                                    uint8_t *h_bwtqs, uint16_t *h_lcp16, uint64_t cap_rows, uint64_t *n_rows,
                                    uint64_t *n_reads)
{
    return guarded(c, [&] {
        fastq_build_ebwt_core(c, TextSrc{HostRef::mem(h_fastq), len}, term_out, HostRef::mem(h_bwt), HostRef::mem(h_bwtqs), HostRef::mem(h_lcp16), 2,
                              cap_rows, n_rows, n_reads);
    });
}

extern "C" int bfq_fastq_build_ebwt_fd(bfq_ctx *c, int fastq_fd, uint64_t len, int term_out, int bwt_fd, int bwtqs_fd, int lcp_fd,
                                       int lcp_bytes, uint64_t *n_rows, uint64_t *n_reads)
{
    return guarded(c, [&] {
        if (fastq_fd < 0) throw BfqError{BFQ_E_ARG, "bad file descriptor"};
        fastq_build_ebwt_oneshot(c, fastq_fd, len, term_out, bwt_fd, bwtqs_fd, lcp_fd, lcp_bytes, n_rows, n_reads);
    });
}

void bfq_fastq_part_index(bfq_ctx *c, const DevFastq *fq, const u64 *h_pstart, int nparts, u64 *d_idx);   // k_fastq.hip
void bfq_pick_u64(bfq_ctx *c, const u64 *d_src, const u64 *d_idx, int count, u64 addIdx, u64 *d_out);      // k_fastq.hip

extern "C" int bfq_fastq_run_job(bfq_ctx *c, bfq_fastq_job *J, bfq_stats *st)
{
    return guarded(c, [&] {
        if (st) memset(st, 0, sizeof *st);
        if (!J || J->nparts < 1 || J->nparts > BFQ_MAX_PARTS || !J->parts) throw BfqError{BFQ_E_ARG, "bfq_fastq_job: 1..BFQ_MAX_PARTS parts"};
        const int np = J->nparts;
        J->fastq_len = J->stream_len = J->hdr_len = J->n_reads = J->total_bases = 0;
        J->dna_bytes = J->qs_bytes = J->hdr_bytes = 0;
        const bool cz = J->compress_streams != 0;
        std::vector<u64> ps;
        TextSrc src[BFQ_MAX_PARTS];
        for (int p = 0; p < np; p++) src[p] = TextSrc{HostRef::mem(J->parts[p].data), J->parts[p].len};
        u8 *d_fq = fastq_upload_and_reserve(c, src, np, ps);
        const u64 len = ps[np];
        c->zeroCounters();
        DevFastq fq;
        bfq_fastq_parse(c, d_fq, len, &fq);
        J->n_reads = fq.N; J->total_bases = fq.total;
        const bool wantStreams = J->out_dna || J->out_qs || J->out_hdr;
        const bool ebwtDomain = J->compress_streams == 2 || J->compress_streams == 3;   // rows of the edited eBWT instead of reads
        const bool qsByRead = J->compress_streams == 3;            // ... but the qualities in read order (they keep their along-the-read correlation)
        if (ebwtDomain && (!J->out_dna || !J->out_qs || J->out_fastq)) throw BfqError{BFQ_E_ARG, "compress_streams = 2 gives out_dna and out_qs (no FASTQ text)"};
        if (c->capped && ebwtDomain) throw BfqError{BFQ_E_NOMEM, "eBWT-domain containers need the LF table: above the workspace cap"};
        const bool lines = J->out_dna || J->out_qs || c->capped;  // the inversion writes the line streams itself (the capped mode knows nothing else)
        const u64 sl = fq.total + fq.N;
        const bool wantLines = J->out_dna || J->out_qs;
        if (wantLines && sl > J->cap_stream) throw BfqError{BFQ_E_ARG, "stream buffer too small (the input length is always enough)"};
        u8 *ob = c->alloc<u8>((lines ? sl : fq.total) + 64), *oq = c->alloc<u8>((lines ? sl : fq.total) + 64);
        u8 *op = ebwtDomain ? c->alloc<u8>(sl + 64) : nullptr;    // eBWT domain: the replaced rows' original symbols
        u8 *lineDna = qsByRead ? c->alloc<u8>(sl + 64) : nullptr, *lineQs = qsByRead ? c->alloc<u8>(sl + 64) : nullptr;
        // per part: index of its first record, and where its share of every output starts (np + 1 entries each)
        u64 *d_pidx = c->alloc<u64>(np + 1), *d_pick = c->alloc<u64>(4 * (np + 1));
        bfq_fastq_part_index(c, &fq, ps.data(), np, d_pidx);
        std::vector<u64> hp(4 * (np + 1), 0);
        bool pickF = false, pickS = false, pickH = false, hdrOnCopyStream = false;
        HIP_CHECK(hipMemcpyAsync(hp.data(), d_pidx, 8 * (np + 1), hipMemcpyDeviceToHost, c->stream));
        if (J->out_hdr) {                                      // the header stream needs only the parsed text: out before the sort starts
            u8 *d_hdr = nullptr;
            u64 *hOff = nullptr;
            u64 hl = 0;
            bfq_fastq_hdr_stream(c, fq.N, d_fq, &fq, &d_hdr, &hl, &hOff);
            J->hdr_len = hl;
            if (cz) {                                          // step 5 on the device: only the container crosses the bus
                const size_t mz = c->mark();
                const u64 bound = bfq_codec_bound(hl) < J->cap_hdr ? bfq_codec_bound(hl) : J->cap_hdr;
                u8 *d_z = c->alloc<u8>(bound + 16);
                J->hdr_bytes = bfq_codec_compress_device(c, d_hdr, hl, d_z, bound);
                bfq_download(c, J->out_hdr, d_z, J->hdr_bytes);
                c->release(mz);
            } else {
                if (hl > J->cap_hdr) throw BfqError{BFQ_E_ARG, "stream buffer too small (the input length is always enough)"};
                if (hl >= (1u << 20) && bfq_is_pinned(J->out_hdr) && !c->env.noOverlap) {
                    // a pinned destination: the copy rides the copy stream beside the sort instead of in front of it
                    if (!c->copyStream) HIP_CHECK(hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
                    hipEvent_t ev;
                    HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                    HIP_CHECK(hipEventRecord(ev, c->stream));
                    HIP_CHECK(hipStreamWaitEvent(c->copyStream, ev, 0));
                    HIP_CHECK(hipMemcpyAsync(J->out_hdr, d_hdr, hl, hipMemcpyDeviceToHost, c->copyStream));
                    (void)hipEventDestroy(ev);
                    hdrOnCopyStream = true;
                } else bfq_download(c, J->out_hdr, d_hdr, hl);
                J->hdr_bytes = hl;
            }
            bfq_pick_u64(c, hOff, d_pidx, np + 1, 0, d_pick + 3 * (np + 1));
            pickH = true;
        }
        if (wantStreams) { bfq_pick_u64(c, fq.roff, d_pidx, np + 1, 1, d_pick + 2 * (np + 1)); pickS = true; }   // roff[i] + i
        size_t m = c->mark();
        StreamOut so{cz ? nullptr : J->out_dna, cz ? nullptr : J->out_qs};
        if (c->capped) {
            steps_capped(c, fq.bases, fq.quals, fq.roff, fq.N, fq.total, ob, oq);
            if (so.h_dna) bfq_download(c, so.h_dna, ob, sl);
            if (so.h_qs) bfq_download(c, so.h_qs, oq, sl);
        } else {
            bfq_step1_device(c, fq.bases, fq.quals, fq.roff, fq.N, fq.total, c->P.term, st);
            EbwtOut eo{ob, oq, op, lineDna, lineQs};                                   // (n = total + N rows: the line-stream buffers have exactly that size)
            steps234_device(c, fq.roff, ob, oq, nullptr, lines ? &so : nullptr, ebwtDomain ? &eo : nullptr);
        }
        if (J->out_dna || J->out_qs) { J->stream_len = sl; if (!cz) { J->dna_bytes = J->out_dna ? sl : 0; J->qs_bytes = J->out_qs ? sl : 0; } }
        c->release(m);                                         // the formatted text may reuse the pipeline's space:
        c->d_bwt = c->d_qual = nullptr; c->d_lcp = nullptr; c->d_gcnt = nullptr;   // the eBWT is gone (bfq_fetch_ebwt refuses)
        if (wantLines && cz) {                                 // step 5 on the device, in the space the pipeline has left
            const size_t mz = c->mark();
            const u64 bound = bfq_codec_bound(sl) < J->cap_stream ? bfq_codec_bound(sl) : J->cap_stream;
            u8 *d_z = c->alloc<u8>(bound + 16);
            if (ebwtDomain) {   // "BFQEBWT1" | rows | reads | terminator byte | 0 | bytes of the symbols' container | that container | the patches' container
                if (bound < 40) throw BfqError{BFQ_E_ARG, "stream buffer too small"};
                const u64 symLen = bfq_codec_compress_device(c, ob, sl, d_z + 40, bound - 40);
                const u64 patLen = bfq_codec_compress_device(c, op, sl, d_z + 40 + symLen, bound - 40 - symLen);
                u8 h[40];
                memcpy(h, "BFQEBWT1", 8);
                const u64 rows = sl, reads = fq.N;
                memcpy(h + 8, &rows, 8); memcpy(h + 16, &reads, 8);
                const u32 tb = (u32)(c->P.term & 0xFF), flags = qsByRead ? 1u : 0u;   // flags bit 0: the quality container is in read order
                memcpy(h + 24, &tb, 4); memcpy(h + 28, &flags, 4); memcpy(h + 32, &symLen, 8);
                HIP_CHECK(hipMemcpyAsync(d_z, h, 40, hipMemcpyHostToDevice, c->stream));
                c->sync();
                J->dna_bytes = 40 + symLen + patLen;
                bfq_download(c, J->out_dna, d_z, J->dna_bytes);
            } else
            if (J->out_dna) { J->dna_bytes = bfq_codec_compress_device(c, ob, sl, d_z, bound); bfq_download(c, J->out_dna, d_z, J->dna_bytes); }
            if (J->out_qs) { J->qs_bytes = bfq_codec_compress_device(c, qsByRead ? lineQs : oq, sl, d_z, bound); bfq_download(c, J->out_qs, d_z, J->qs_bytes); }
            c->release(mz);
        }
        if (J->out_fastq) {
            u8 *d_out = nullptr;
            u64 *recOff = nullptr;
            u64 ol = bfq_fastq_format(c, ob, oq, fq.roff, fq.N, J->keep_headers ? 2 : 0, d_fq, len, &fq, &d_out, &recOff, lines);
            J->fastq_len = ol;
            if (ol > J->cap_fastq) throw BfqError{BFQ_E_ARG, "output buffer smaller than the FASTQ text (see bfq_fastq_out_bound)"};
            bfq_download(c, J->out_fastq, d_out, ol);
            bfq_pick_u64(c, recOff, d_pidx, np + 1, 0, d_pick + (np + 1));
            pickF = true;
        }
        HIP_CHECK(hipMemcpyAsync(hp.data() + (np + 1), d_pick + (np + 1), 8 * 3 * (np + 1), hipMemcpyDeviceToHost, c->stream));
        if (hdrOnCopyStream) HIP_CHECK(hipStreamSynchronize(c->copyStream));
        c->fetchCounters();
        c->profCollect();
        check_counters(c);
        fill_stats(c, st);
        for (int p = 0; p <= np; p++) {
            J->part_reads[p] = hp[p];
            J->part_fastq_off[p] = pickF ? hp[(np + 1) + p] : 0;
            J->part_stream_off[p] = pickS ? hp[2 * (np + 1) + p] : 0;
            J->part_hdr_off[p] = pickH ? hp[3 * (np + 1) + p] : 0;
        }
    });
}

extern "C" int bfq_fastq_run(bfq_ctx *c, const uint8_t *h_fastq, uint64_t len, int keep_headers, uint8_t *h_out,
                             uint64_t cap, uint64_t *out_len, bfq_stats *st)
{
    bfq_text_part part{h_fastq, len};
    bfq_fastq_job J;
    memset(&J, 0, sizeof J);
    J.parts = &part; J.nparts = 1; J.keep_headers = keep_headers;
    J.out_fastq = h_out; J.cap_fastq = cap;
    if (!h_out) return BFQ_E_ARG;
    int rc = bfq_fastq_run_job(c, &J, st);
    if (out_len) *out_len = J.fastq_len;
    return rc;
}

extern "C" int bfq_fastq_run_streams(bfq_ctx *c, const uint8_t *h_fastq, uint64_t len, uint8_t *h_dna, uint8_t *h_qs,
                                     uint64_t cap_stream, uint64_t *stream_len, uint8_t *h_hdr, uint64_t cap_hdr,
                                     uint64_t *hdr_len, bfq_stats *st)
{
    bfq_text_part part{h_fastq, len};
    bfq_fastq_job J;
    memset(&J, 0, sizeof J);
    J.parts = &part; J.nparts = 1;
    J.out_dna = h_dna; J.out_qs = h_qs; J.cap_stream = cap_stream;
    J.out_hdr = h_hdr; J.cap_hdr = cap_hdr;
    if (!h_dna && !h_qs && !h_hdr) return BFQ_E_ARG;
    int rc = bfq_fastq_run_job(c, &J, st);
    if (stream_len) *stream_len = J.stream_len;
    if (hdr_len) *hdr_len = J.hdr_len;
    return rc;
}

static void smooth_invert_fastq_core(bfq_ctx *c, HostRef bwt, HostRef qs, HostRef lcp, int lcp_bytes, uint64_t n, HostRef headers,
                                     bool haveHeaders, uint64_t headers_len, HostRef out, uint64_t cap, uint64_t *out_len, bfq_stats *st,
                                     const OutFile *outFile = nullptr)
{
    SmoothOut r;
    smooth_invert_core(c, bwt, qs, lcp, lcp_bytes, n, 3 * (n + 4096) + 2 * headers_len + (32u << 20), st, &r);
    if (outFile && outFile->m) bfq_outmap_extend(outFile->m, haveHeaders ? headers_len + 2 * r.total + 4 * r.N : 2 * r.total + 6 * r.N);
    u8 *d_hdr = nullptr;
    if (haveHeaders) {
        d_hdr = c->alloc<u8>(headers_len + 64);
        bfq_upload(c, d_hdr, headers, headers_len);
    }
    u8 *d_out = nullptr;
    u64 ol = bfq_fastq_format(c, r.ob, r.oq, r.roff, r.N, haveHeaders ? 1 : 0, d_hdr, headers_len, nullptr, &d_out);
    if (out_len) *out_len = ol;
    if (ol > cap) throw BfqError{BFQ_E_ARG, "output buffer smaller than the FASTQ text (see bfq_fastq_out_bound)"};
    if (outFile) {
        if (outFile->m && ol > bfq_outmap_len(outFile->m)) throw BfqError{BFQ_E_IO, "output mapping smaller than the FASTQ text"};
        bfq_write_async(c, outFile->at(0), d_out, ol);
        c->fetchCounters();
        c->profCollect();
        bfq_phase("d2h_write");
        bfq_write_wait(c);
    } else {
        bfq_download(c, out, d_out, ol);
        c->fetchCounters();
        c->profCollect();
    }
    check_counters(c);
    fill_stats(c, st);
}

extern "C" int bfq_smooth_invert_fastq(bfq_ctx *c, const uint8_t *h_bwt, const uint8_t *h_bwtqs, const void *h_lcp,
                                       int lcp_bytes, uint64_t n, const uint8_t *h_headers, uint64_t headers_len,
                                       uint8_t *h_out, uint64_t cap, uint64_t *out_len, bfq_stats *st)
{
    return guarded(c, [&] {
        smooth_invert_fastq_core(c, HostRef::mem(h_bwt), HostRef::mem(h_bwtqs), HostRef::mem(h_lcp), lcp_bytes, n, HostRef::mem(h_headers),
                                 h_headers != nullptr, headers_len, HostRef::mem(h_out), cap, out_len, st);
    });
}

extern "C" int bfq_smooth_invert_fastq_fd(bfq_ctx *c, int bwt_fd, int qs_fd, int lcp_fd, int lcp_bytes, uint64_t n, int headers_fd,
                                          uint64_t headers_len, int out_fd, uint64_t *out_len, bfq_stats *st)
{
    return guarded(c, [&] {
        if (bwt_fd < 0 || qs_fd < 0 || out_fd < 0) throw BfqError{BFQ_E_ARG, "bad file descriptor"};
        // the FASTQ text is at least 2 n bytes (bases + qualities + their newlines) and at most 6 n + the header file
        // (a collection of empty reads): mapped to the bound, pre-faulted to what is certain while the eBWT is uploaded
        OutFile of;
        of.open(out_fd, 6 * n + headers_len + 4096, 2 * n);
        uint64_t ol = 0;
        try {
            smooth_invert_fastq_core(c, HostRef::file(bwt_fd), HostRef::file(qs_fd), lcp_fd >= 0 ? HostRef::file(lcp_fd) : HostRef(), lcp_bytes, n,
                                     headers_fd >= 0 ? HostRef::file(headers_fd) : HostRef(), headers_fd >= 0, headers_len, HostRef::file(out_fd),
                                     ~0ull, &ol, st, &of);
        } catch (...) {
            try { bfq_write_wait(c); } catch (...) {}
            of.close(0);
            throw;
        }
        if (out_len) *out_len = ol;
        if (!of.close(ol)) throw BfqError{BFQ_E_IO, "cannot size the output file"};
    });
}

// ---------------------------------------------------------------- synthetic reads
extern "C" void bfq_synth_default(bfq_synth *s, uint64_t N, uint32_t L)
{
    memset(s, 0, sizeof *s);
    s->seed = 20240807; s->N = N; s->Lmin = s->Lmax = L; s->coverage = 30;
    s->err_ppm = 10000; s->n_ppm = 1000; s->snp_every = 1000; s->dsnp_every = 10000; s->both_strands = 1;
}
extern "C" uint64_t bfq_synth_total(const bfq_synth *s)
{
    if (s->Lmin >= s->Lmax) return s->N * (u64)s->Lmin;
    u64 t = 0;
    for (u64 i = 0; i < s->N; i++) t += bfq_synth_len(s, s->first + i);
    return t;
}
extern "C" int bfq_synth_host(const bfq_synth *s, uint8_t *h_bases, uint8_t *h_quals, uint64_t *h_read_off)
{
    if (!s || !h_read_off) return BFQ_E_ARG;
    u64 o = 0;
    for (u64 i = 0; i < s->N; i++) {
        u32 len = bfq_synth_len(s, s->first + i);
        h_read_off[i] = o;
        for (u32 k = 0; k < len; k++) bfq_synth_base(s, s->first + i, len, k, h_bases + o + k, h_quals + o + k);
        o += len;
    }
    h_read_off[s->N] = o;
    return BFQ_OK;
}
extern "C" int bfq_synth_device(bfq_ctx *c, const bfq_synth *s, uint8_t *d_bases, uint8_t *d_quals, uint64_t *d_read_off)
{
    return guarded(c, [&] {
        if (!s) throw BfqError{BFQ_E_ARG, "null synth spec"};
        c->reserve(16 * (s->N + 4096) + (64u << 20));
        bfq_synth_launch(c, s, d_bases, d_quals, (u64 *)d_read_off);
        c->sync();
        c->profCollect();
    });
}

u8 *bfq_synth_headers(bfq_ctx *c, const bfq_synth *s, u64 *len);   // k_synth.hip
extern "C" int bfq_synth_fastq(bfq_ctx *c, const bfq_synth *s, uint8_t *h_out, uint64_t cap, uint64_t *out_len)
{
    return guarded(c, [&] {
        if (!s || !h_out) throw BfqError{BFQ_E_ARG, "null synth spec / output"};
        const u64 maxTotal = s->N * (u64)s->Lmax;
        c->reserve(4 * (maxTotal + 4096) + 96 * (s->N + 4096) + (64u << 20));
        u8 *db = c->alloc<u8>(maxTotal + 64), *dq = c->alloc<u8>(maxTotal + 64);
        u64 *dr = c->alloc<u64>(s->N + 2);
        bfq_synth_launch(c, s, db, dq, dr);
        u64 hl = 0;
        u8 *hdr = bfq_synth_headers(c, s, &hl);
        u8 *d_out = nullptr;
        u64 ol = bfq_fastq_format(c, db, dq, dr, s->N, 1, hdr, hl, nullptr, &d_out);
        if (out_len) *out_len = ol;
        if (ol > cap) throw BfqError{BFQ_E_ARG, "output buffer smaller than the FASTQ text"};
        bfq_download(c, h_out, d_out, ol);
        c->sync();
        c->profCollect();
    });
}

// ---------------------------------------------------------------- stream codec (BFQzip.py:253-275)
extern "C" uint64_t bfq_stream_bound(uint64_t len) { return bfq_codec_bound(len); }
extern "C" int64_t bfq_stream_raw_len(const uint8_t *h_in, uint64_t len)
{
    try { return (int64_t)bfq_codec_raw_len(h_in, len); } catch (const BfqError &) { return -1; }
}
extern "C" int bfq_stream_compress(bfq_ctx *c, const uint8_t *h_in, uint64_t len, uint8_t *h_out, uint64_t cap, uint64_t *out_len)
{
    return guarded(c, [&] {
        if ((len && !h_in) || !h_out || !out_len) throw BfqError{BFQ_E_ARG, "null argument"};
        c->reserve(bfq_codec_workspace(len));
        u8 *d_in = c->alloc<u8>(len + 16);
        const u64 bound = bfq_codec_bound(len) < cap ? bfq_codec_bound(len) : cap;
        u8 *d_out = c->alloc<u8>(bound + 16);
        if (len) bfq_upload(c, d_in, h_in, len);
        const u64 got = bfq_codec_compress_device(c, d_in, len, d_out, bound);
        bfq_download(c, h_out, d_out, got);
        c->sync();                                             // pinned destinations are written by asynchronous DMA
        c->profCollect();
        *out_len = got;
    });
}
extern "C" int bfq_stream_decompress(bfq_ctx *c, const uint8_t *h_in, uint64_t len, uint8_t *h_out, uint64_t cap, uint64_t *out_len)
{
    return guarded(c, [&] {
        if (!h_in || !out_len) throw BfqError{BFQ_E_ARG, "null argument"};
        const u64 raw = bfq_codec_raw_len(h_in, len);
        if (raw > cap || (raw && !h_out)) throw BfqError{BFQ_E_ARG, "output buffer too small for the raw stream"};
        c->reserve(bfq_codec_workspace(raw) + len);
        u8 *d_in = c->alloc<u8>(len + 16), *d_out = c->alloc<u8>(raw + 16);
        bfq_upload(c, d_in, h_in, len);
        u64 got = 0;
        for (u64 pos = 0; pos < len;) {                            // one container per block of a sharded run, back to back
            const u64 ml = bfq_codec_member_len(h_in + pos, len - pos);
            got += bfq_codec_decompress_device(c, h_in + pos, d_in + pos, ml, d_out + got, raw - got);
            pos += ml;
        }
        if (got) bfq_download(c, h_out, d_out, got);
        c->sync();
        c->profCollect();
        *out_len = got;
    });
}
// both buffers on the device (what bench.py times); the workspace is reserved by bfq_stream_reserve first
extern "C" int bfq_stream_reserve(bfq_ctx *c, uint64_t len)
{
    return guarded(c, [&] { c->reserve(bfq_codec_workspace(len)); });
}
extern "C" int bfq_stream_compress_device(bfq_ctx *c, const uint8_t *d_in, uint64_t len, uint8_t *d_out, uint64_t cap, uint64_t *out_len)
{
    return guarded(c, [&] {
        c->wsTop = 0;
        *out_len = bfq_codec_compress_device(c, d_in, len, d_out, cap);
        c->sync();
        c->profCollect();
    });
}

// orig[r] = patch[r] ? patch[r] : sym[r] (in place in `orig`, which arrives holding the patches)
__global__ __launch_bounds__(256) void k_ebwt_unpatch(const u8 *__restrict__ sym, u8 *__restrict__ orig, u64 n)
{
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x)
        if (!orig[r]) orig[r] = sym[r];
}
// rows whose symbol was replaced: flag + replacement in the LF entry, as k_cluster leaves them
__global__ __launch_bounds__(256) void k_ebwt_mark(const u8 *__restrict__ sym, const u8 *__restrict__ orig, u64 n, u64 *__restrict__ lfq)
{
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x)
        if (sym[r] != orig[r]) lfq_set_repl(lfq, r, bfq_base_code(orig[r]), bfq_base_code(sym[r]));
}
// eBWT-domain containers (bfq_fastq_job.compress_streams = 2) back to the line streams OUT.fq.dna / OUT.fq.qs: the two
// containers are decoded on the device, the LF table is built from the rows and the reads are walked out (steps 4 of the
// path; no clusters: the rows already hold the smoothed result).
extern "C" int bfq_stream_ebwt_decode(bfq_ctx *c, const uint8_t *h_bwtz, uint64_t len_b, const uint8_t *h_qsz, uint64_t len_q,
                                      uint8_t *h_dna, uint8_t *h_qs, uint64_t cap, uint64_t *stream_len, uint64_t *n_reads)
{
    return guarded(c, [&] {
        if (!h_bwtz || !h_qsz || len_b < 40 || memcmp(h_bwtz, "BFQEBWT1", 8)) throw BfqError{BFQ_E_ARG, "not a BFQEBWT1 stream"};
        u64 n = 0, N = 0, symLen = 0;
        u32 tb = 0, flags = 0;
        memcpy(&n, h_bwtz + 8, 8); memcpy(&N, h_bwtz + 16, 8); memcpy(&tb, h_bwtz + 24, 4); memcpy(&flags, h_bwtz + 28, 4); memcpy(&symLen, h_bwtz + 32, 8);
        const bool qsByRead = flags & 1u;
        const BfqError bad{BFQ_E_ARG, "damaged BFQEBWT1 stream"};
        if (symLen > len_b - 40 || N > n) throw bad;
        const u8 *h_sym = h_bwtz + 40, *h_pat = h_bwtz + 40 + symLen;
        const u64 patLen = len_b - 40 - symLen;
        if (bfq_codec_raw_len(h_sym, symLen) != n || bfq_codec_raw_len(h_pat, patLen) != n || bfq_codec_raw_len(h_qsz, len_q) != n) throw bad;
        if (n > cap) throw BfqError{BFQ_E_ARG, "output buffer too small for the streams"};
        if (stream_len) *stream_len = n;
        if (n_reads) *n_reads = N;
        if (!n) return;
        c->reserve(ws_need_given(n, N, 5 * n + len_b + len_q + bfq_codec_workspace(n) / 2 + (64u << 20)));
        c->zeroCounters();
        c->n = n; c->N = N;
        c->d_bwt = c->alloc<u8>(n + 64); c->d_qual = c->alloc<u8>(n + 64);
        u8 *d_sym = c->alloc<u8>(n + 64);                          // the symbols the reads get; d_bwt = the eBWT the walk navigates by
        {
            const size_t mk = c->mark();
            u8 *d_z = c->alloc<u8>((len_b > len_q ? len_b : len_q) + 64);
            bfq_upload(c, d_z, h_sym, symLen);
            bfq_codec_decompress_device(c, h_sym, d_z, symLen, d_sym, n);
            bfq_upload(c, d_z, h_pat, patLen);
            bfq_codec_decompress_device(c, h_pat, d_z, patLen, c->d_bwt, n);
            KLAUNCH(c, K_MISC, 3.0 * (double)n, k_ebwt_unpatch, bfq_grid(n, 256 * 16), 256, (const u8 *)d_sym, c->d_bwt, n);
            bfq_upload(c, d_z, h_qsz, len_q);
            bfq_codec_decompress_device(c, h_qsz, d_z, len_q, c->d_qual, n);
            if (qsByRead) {                                        // already the line stream OUT.fq.qs: out as it is; the walk carries dummies
                if (h_qs) bfq_download(c, h_qs, c->d_qual, n);
                HIP_CHECK(hipMemsetAsync(c->d_qual, '!', n, c->stream));
            }
            c->release(mk);
        }
        u64 *d_roff = c->alloc<u64>(N + 2);
        u32 *lens = c->alloc<u32>(N + 2);
        u8 *d_dna = c->alloc<u8>(n + 64), *d_q = c->alloc<u8>(n + 64);
        const bfq_params keep = c->P;
        c->P.term = (int)tb; c->P.B = 0;                           // the rows are binned already
        try {
            RankIndex R = bfq_rank_build(c, c->d_bwt, c->d_qual, n, c->P.term, nullptr);
            KLAUNCH(c, K_MISC, 2.0 * (double)n, k_ebwt_mark, bfq_grid(n, 256 * 16), 256, (const u8 *)d_sym, (const u8 *)c->d_bwt, n, R.lfq);
            c->fetchCounters();
            if (c->h_cnt.tot[0] != N) throw BfqError{BFQ_E_NOT_EBWT, "terminator rows do not match the header"};
            bool guessed = false;
            if (N && (n - N) % N == 0) { bfq_fixed_offsets(c, N, (n - N) / N, d_roff); guessed = true; }
            else count_lengths(c, R, d_roff, lens);
            StreamOut so{h_dna, qsByRead ? nullptr : h_qs};
            invert_lines(c, R, d_roff, d_dna, d_q, &so);
            c->fetchCounters();
            if (guessed && c->h_cnt.errInvert) {                   // not all of one length after all
                HIP_CHECK(hipMemsetAsync(&c->d_cnt->errInvert, 0, sizeof(u64), c->stream));
                count_lengths(c, R, d_roff, lens);
                invert_lines(c, R, d_roff, d_dna, d_q, &so);
                c->fetchCounters();
            }
            c->P = keep;
        } catch (...) { c->P = keep; throw; }
        c->profCollect();
        check_counters(c);
        c->d_bwt = c->d_qual = nullptr;
    });
}
