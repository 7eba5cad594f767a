// bfq_io.hip -- host <-> device transfers of the FASTQ entry points and the host-side line index.
//
// The reference moves its data through files and pipes (`sed -n 2~4p`, `cat`, Python line
// loops: BFQzip.py:192-251, BFQzip_parallel.py:288-360,137-179).  Here a block of FASTQ text
// goes to the GPU as bytes and comes back as bytes, so the transfer itself is what has to be
// fast:
//   * a host buffer that is pinned (bfq_host_alloc / hipHostMalloc / hipHostRegister) is
//     copied by one asynchronous DMA on the context's stream;
//   * a pageable buffer (an mmap'ed file, a std::vector) is staged: worker threads copy
//     chunks into their own pinned staging buffers and DMA them on their own streams, two
//     buffers per worker, so that the CPU copy of one chunk overlaps the DMA of the other.
// Host-side line index (bfq_text_count_lines / bfq_text_nth_newline): what the multi-GPU
// driver needs to cut a file into the blocks of BFQzip_parallel.split_fastq without
// parsing it (one sharded pass over the bytes).
#include <string.h>
#include <stdlib.h>
#include <unistd.h>
#include <sys/mman.h>
#include <errno.h>
#include <time.h>
#include <stdio.h>
#include <thread>
#include <algorithm>
#include <atomic>
#include "bfq_internal.h"

static int io_threads()
{
    static const int t = [] {
        { int v = bfq_env().ioThreads; if (v >= 1 && v <= BFQ_IO_MAX_WORKERS) return v; }
        // staging workers beside the main thread, the threads that prepare the output files (one fallocate + two populate
        // per tool) and the runtime's own: 6 of a budget of 16.  6 workers move 30-38 GB/s when they have their CPUs; 12
        // (three quarters of a cgroup quota of 16) put the process over its quota together with the helpers, and every
        // phase paid: tools at 30 M x 150, gsufsort / bfq_int wall, three runs each -- 12 workers 2.77 3.17 2.77 / 5.62 3.82
        // 3.00 s, 6 workers 2.73 2.87 2.36 / 2.75 3.16 2.33 s (profiles/r3/dropin_phases.md)
        const int cpus = bfq_cpu_budget();
        return cpus >= 16 ? 6 : cpus >= 8 ? 4 : cpus >= 4 ? 2 : 1;
    }();
    return t;
}

bool bfq_is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

extern "C" void *bfq_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void bfq_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

void bfq_ctx::ioInit(int want)
{
    const int T = std::min(io_threads(), std::max(want, 1));
    for (int t = ioWorkers; t < T; t++) {                       // grows on demand, never shrinks
        IoWorker &w = io[t];
        HIP_CHECK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) HIP_CHECK(hipHostMalloc((void **)&w.stage[k], BFQ_IO_STAGE_BYTES, hipHostMallocDefault));
        ioWorkers = t + 1;
    }
}
static void writer_free(bfq_ctx *c);
void bfq_ctx::ioFree()
{
    writer_free(this);
    for (int t = 0; t < BFQ_IO_MAX_WORKERS; t++) {
        IoWorker &w = io[t];
        if (w.stream) { (void)hipStreamSynchronize(w.stream); (void)hipStreamDestroy(w.stream); w.stream = nullptr; }
        for (int k = 0; k < 2; k++) {
            if (w.stage[k]) { (void)hipHostFree(w.stage[k]); w.stage[k] = nullptr; }
            if (w.done[k]) { (void)hipEventDestroy(w.done[k]); w.done[k] = nullptr; }
        }
    }
    ioWorkers = 0;
}

// host side of a staged chunk: memory, or a file at an offset (pread / pwrite: the kernel copies between the page
// cache and the pinned staging buffer, no mapping, no page faults in user space)
static bool host_get(const HostRef &h, size_t off, char *dst, size_t sz)
{
    if (h.ptr) { memcpy(dst, (const char *)h.ptr + off, sz); return true; }
    size_t got = 0;
    while (got < sz) {
        ssize_t r = pread(h.fd, dst + got, sz - got, (off_t)(h.off + off + got));
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return false;
        got += (size_t)r;
    }
    return true;
}
static bool host_put(const HostRef &h, size_t off, const char *src, size_t sz)
{
    if (h.ptr) {
        if (h.om && !bfq_outmap_ensure(h.om, h.off + off, sz)) {    // no pages to be had (disk full ...): pwrite reports it
            HostRef f = HostRef::file(bfq_outmap_fd(h.om), h.off);
            return host_put(f, off, src, sz);
        }
        memcpy((char *)h.ptr + off, src, sz);
        return true;
    }
    size_t put = 0;
    while (put < sz) {
        ssize_t w = pwrite(h.fd, src + put, sz - put, (off_t)(h.off + off + put));
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) return false;
        put += (size_t)w;
    }
    return true;
}

// Chunk i of the transfer belongs to worker i mod T.  `up`: host -> device, else device -> host.
static bool io_trace() { return bfq_env().trace; }
static double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

extern std::atomic<int> g_bfqUploadsRunning;                   // bfq_host.cpp: the output helpers stand back meanwhile
static void staged_copy(bfq_ctx *c, char *dev, HostRef host, size_t len, bool up)
{
    struct Busy { bool on; Busy(bool o) : on(o) { if (on) g_bfqUploadsRunning++; } ~Busy() { if (on) g_bfqUploadsRunning--; } } busy(up && !host.ptr);
    const double t0 = io_trace() ? now_s() : 0;
    // one worker per 128 MiB (each costs two pinned 16 MiB buffers the first time): a 200 MB file is not worth sixteen of them
    const int want = (int)std::min<size_t>(BFQ_IO_MAX_WORKERS, (len + (128u << 20) - 1) / (128u << 20));
    c->ioInit(want);
    const int T = std::min(c->ioWorkers, std::max(want, 1));
    const size_t CH = BFQ_IO_STAGE_BYTES;
    const size_t nch = (len + CH - 1) / CH;
    hipError_t errs[BFQ_IO_MAX_WORKERS];
    for (int t = 0; t < T; t++) errs[t] = hipSuccess;
    std::atomic<bool> ioFail{false};
    auto work = [&](int t) {
        bfq_ctx::IoWorker &w = c->io[t];
        hipError_t e = hipSetDevice(c->device);
        size_t pend[2] = {0, 0}, poff[2] = {0, 0};              // download: chunk waiting in each staging buffer
        int k = 0;
        for (size_t i = (size_t)t; i < nch && e == hipSuccess; i += (size_t)T, k ^= 1) {
            const size_t off = i * CH, sz = std::min(CH, len - off);
            if (up) {
                // stage[k] was handed to the DMA two chunks ago: the stream is in order, so waiting for the
                // previous chunk's DMA (issued from stage[k^1]) is not needed -- only stage[k]'s own
                if (i >= (size_t)(2 * T)) e = hipEventSynchronize(w.done[k]);
                if (e != hipSuccess) break;
                if (!host_get(host, off, w.stage[k], sz)) { ioFail = true; break; }
                e = hipMemcpyAsync(dev + off, w.stage[k], sz, hipMemcpyHostToDevice, w.stream);
                if (e == hipSuccess) e = hipEventRecord(w.done[k], w.stream);
            } else {
                e = hipMemcpyAsync(w.stage[k], dev + off, sz, hipMemcpyDeviceToHost, w.stream);
                if (e == hipSuccess) e = hipEventRecord(w.done[k], w.stream);
                if (pend[k ^ 1]) {                               // while that DMA runs: drain the other buffer
                    if (e == hipSuccess) e = hipEventSynchronize(w.done[k ^ 1]);
                    if (e == hipSuccess && !host_put(host, poff[k ^ 1], w.stage[k ^ 1], pend[k ^ 1])) ioFail = true;
                    pend[k ^ 1] = 0;
                }
                pend[k] = sz; poff[k] = off;
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(w.stream);
        if (!up && e == hipSuccess)
            for (int q = 0; q < 2; q++)
                if (pend[q] && !host_put(host, poff[q], w.stage[q], pend[q])) ioFail = true;
        errs[t] = e;
    };
    for (int t = 0; t < T; t++)
        for (int k = 0; k < 2; k++)
            if (!c->io[t].done[k]) HIP_CHECK(hipEventCreateWithFlags(&c->io[t].done[k], hipEventDisableTiming));
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (int t = 0; t < T; t++)
        if (errs[t] != hipSuccess) throw BfqError{BFQ_E_HIP, std::string("staged transfer: ") + hipGetErrorString(errs[t])};
    if (ioFail) throw BfqError{BFQ_E_IO, std::string("staged transfer: file read / write failed: ") + strerror(errno)};
    if (io_trace()) {
        const double dt = now_s() - t0;
        fprintf(stderr, "[bfq io] %s %s %.2f GB in %.3f s = %.1f GB/s (%d workers)\n", up ? "upload" : "download", host.ptr ? "memory" : "file",
                len / 1e9, dt, len / 1e9 / dt, T);
    }
}

// host -> device.  Ordered after everything already on the context's stream; when it returns the
// data is either on the device (staged path) or queued on the context's stream (pinned source).
void bfq_upload(bfq_ctx *c, void *d_dst, HostRef src, size_t len)
{
    if (!len) return;
    if (src.ptr && (bfq_is_pinned(src.ptr) || len < (1u << 20))) {
        HIP_CHECK(hipMemcpyAsync(d_dst, src.ptr, len, hipMemcpyHostToDevice, c->stream));
        return;
    }
    c->sync();                                          // d_dst may still be in use by queued kernels
    staged_copy(c, (char *)d_dst, src, len, true);
}
void bfq_upload(bfq_ctx *c, void *d_dst, const void *h_src, size_t len) { bfq_upload(c, d_dst, HostRef::mem(h_src), len); }

// device -> host, same rules; with a pageable destination the call returns with the bytes in place
void bfq_download(bfq_ctx *c, HostRef dst, const void *d_src, size_t len)
{
    if (!len) return;
    if (dst.ptr && (bfq_is_pinned(dst.ptr) || len < (1u << 20))) {
        HIP_CHECK(hipMemcpyAsync(dst.ptr, d_src, len, hipMemcpyDeviceToHost, c->stream));
        return;
    }
    c->sync();                                          // the producer kernels run on the context's stream
    if (!dst.ptr) {
        // A file: buffered pwrite()s of one file serialise on its inode lock (3.5 GB/s measured on tmpfs with 8 writers);
        // page faults on a shared mapping do not (6.4 GB/s), so the range is mapped when the descriptor allows it.
        const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
        const u64 a0 = dst.off / pg * pg;
        if (ftruncate(dst.fd, (off_t)(dst.off + len)) == 0) {
            void *m = mmap(nullptr, (size_t)(dst.off + len - a0), PROT_READ | PROT_WRITE, MAP_SHARED, dst.fd, (off_t)a0);
            if (m != MAP_FAILED) {
                try { staged_copy(c, (char *)d_src, HostRef::mem((char *)m + (dst.off - a0)), len, false); }
                catch (...) { munmap(m, (size_t)(dst.off + len - a0)); throw; }
                munmap(m, (size_t)(dst.off + len - a0));
                return;
            }
        }
    }
    staged_copy(c, (char *)d_src, dst, len, false);
}
void bfq_download(bfq_ctx *c, void *h_dst, const void *d_src, size_t len) { bfq_download(c, HostRef::mem(h_dst), d_src, len); }

// ---------------------------------------------------------------- background writes (one-shot tools)
// The drop-in tools write 9 GB files (BFQzip.py:184,215-222 at 30 M x 150 bp).  Their outputs leave the device while the GPU
// is still working on the next pile / chunk: a few writer threads, each with its own stream and two pinned staging buffers,
// take 16 MiB pieces from a queue -- DMA into one buffer while the other is copied into the (pre-faulted) output mapping.
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
struct WrTask { HostRef dst; const char *src; size_t sz; hipEvent_t after; };
struct BfqWriter {
    bfq_ctx *c = nullptr;
    bfq_ctx::IoWorker w[BFQ_IO_MAX_WORKERS];
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cvIdle;
    std::deque<WrTask> q;
    size_t inflight = 0;                                        // queued or staged, not yet at their destination
    bool quit = false;
    hipError_t err = hipSuccess;
    bool ioFail = false;
    std::vector<hipEvent_t> events;
    double bytes = 0, t0 = 0;
};
static void writer_thread(BfqWriter *W, int t)
{
    bfq_ctx::IoWorker &w = W->w[t];
    hipError_t e = hipSetDevice(W->c->device);
    WrTask pend[2];
    pend[0].sz = pend[1].sz = 0;
    auto flush = [&](int k) {
        if (!pend[k].sz) return;
        hipError_t fe = hipEventSynchronize(w.done[k]);
        bool bad = false;
        if (fe == hipSuccess) bad = !host_put(pend[k].dst, 0, w.stage[k], pend[k].sz);
        pend[k].sz = 0;
        std::lock_guard<std::mutex> g(W->mu);
        if (fe != hipSuccess && W->err == hipSuccess) W->err = fe;
        if (bad) W->ioFail = true;
        W->inflight--;
        W->cvIdle.notify_all();
    };
    int k = 0;
    for (;;) {
        WrTask task;
        {
            std::unique_lock<std::mutex> lk(W->mu);
            if (W->q.empty() && (pend[0].sz || pend[1].sz)) { lk.unlock(); flush(k ^ 1); flush(k); continue; }
            W->cv.wait(lk, [&] { return W->quit || !W->q.empty(); });
            if (W->q.empty()) break;                            // quit
            task = W->q.front();
            W->q.pop_front();
        }
        flush(k);                                               // stage[k] is free again
        if (e == hipSuccess && task.after) e = hipStreamWaitEvent(w.stream, task.after, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(w.stage[k], task.src, task.sz, hipMemcpyDeviceToHost, w.stream);
        if (e == hipSuccess) e = hipEventRecord(w.done[k], w.stream);
        if (e != hipSuccess) {
            std::lock_guard<std::mutex> g(W->mu);
            if (W->err == hipSuccess) W->err = e;
            W->inflight--;
            W->cvIdle.notify_all();
            e = hipSuccess;
            (void)hipGetLastError();
            continue;
        }
        pend[k] = task;
        flush(k ^ 1);                                           // the other buffer drains while this DMA runs
        k ^= 1;
    }
    flush(0); flush(1);
}
static BfqWriter *writer_get(bfq_ctx *c, size_t firstBytes)
{
    if (c->writer) return c->writer;
    BfqWriter *W = new BfqWriter();
    W->c = c;
    // sized by what the caller expects to write in all (bfq_ctx::writeHint) or by the first job: one thread per 128 MiB
    const int T = (int)std::min<size_t>((size_t)io_threads(), std::max<size_t>(1, (firstBytes + (128u << 20) - 1) / (128u << 20)));
    for (int t = 0; t < T; t++) {
        HIP_CHECK(hipStreamCreateWithFlags(&W->w[t].stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            HIP_CHECK(hipHostMalloc((void **)&W->w[t].stage[k], BFQ_IO_STAGE_BYTES, hipHostMallocDefault));
            HIP_CHECK(hipEventCreateWithFlags(&W->w[t].done[k], hipEventDisableTiming));
        }
    }
    c->writer = W;
    for (int t = 0; t < T; t++) W->th.emplace_back(writer_thread, W, t);
    return W;
}
void bfq_write_async(bfq_ctx *c, HostRef dst, const void *d_src, size_t len)
{
    if (!len) return;
    BfqWriter *W = writer_get(c, std::max<size_t>(len, c->writeHint));
    hipEvent_t ev;
    HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ev, c->stream));
    const size_t CH = BFQ_IO_STAGE_BYTES;
    std::lock_guard<std::mutex> g(W->mu);
    if (!W->inflight && W->bytes == 0) W->t0 = now_s();
    W->events.push_back(ev);
    for (size_t off = 0; off < len; off += CH) {
        WrTask t;
        t.dst = dst;
        if (dst.ptr) t.dst.ptr = (char *)dst.ptr + off;
        t.dst.off = dst.off + off;
        t.src = (const char *)d_src + off;
        t.sz = std::min(CH, len - off);
        t.after = ev;
        W->q.push_back(t);
        W->inflight++;
    }
    W->bytes += (double)len;
    W->cv.notify_all();
}
void bfq_write_wait(bfq_ctx *c)
{
    BfqWriter *W = c->writer;
    if (!W) return;
    hipError_t e;
    bool bad;
    {
        std::unique_lock<std::mutex> lk(W->mu);
        W->cvIdle.wait(lk, [&] { return W->inflight == 0; });
        e = W->err; bad = W->ioFail;
        W->err = hipSuccess; W->ioFail = false;
        for (auto ev : W->events) (void)hipEventDestroy(ev);
        W->events.clear();
        if (io_trace() && W->bytes > 0)
            fprintf(stderr, "[bfq io] background writes: %.2f GB, last byte %.3f s after the first was queued\n", W->bytes / 1e9, now_s() - W->t0);
        W->bytes = 0;
    }
    if (e != hipSuccess) throw BfqError{BFQ_E_HIP, std::string("background write: ") + hipGetErrorString(e)};
    if (bad) throw BfqError{BFQ_E_IO, std::string("background write: file write failed: ") + strerror(errno)};
}
static void writer_free(bfq_ctx *c)
{
    BfqWriter *W = c->writer;
    if (!W) return;
    {
        std::lock_guard<std::mutex> g(W->mu);
        W->quit = true;
        W->cv.notify_all();
    }
    for (auto &t : W->th) t.join();
    for (auto ev : W->events) (void)hipEventDestroy(ev);
    for (int t = 0; t < BFQ_IO_MAX_WORKERS; t++) {
        bfq_ctx::IoWorker &w = W->w[t];
        if (w.stream) { (void)hipStreamSynchronize(w.stream); (void)hipStreamDestroy(w.stream); }
        for (int k = 0; k < 2; k++) {
            if (w.stage[k]) (void)hipHostFree(w.stage[k]);
            if (w.done[k]) (void)hipEventDestroy(w.done[k]);
        }
    }
    delete W;
    c->writer = nullptr;
}

// upload on a helper thread (the staging workers do the copying; the caller's thread stays free to launch kernels)
struct BfqAsyncUpload { std::thread th; std::exception_ptr ex; BfqError err{0, ""}; bool failed = false; };
BfqAsyncUpload *bfq_upload_begin(bfq_ctx *c, void *d_dst, HostRef src, size_t len)
{
    BfqAsyncUpload *u = new BfqAsyncUpload();
    if (!len) return u;
    c->ioInit(BFQ_IO_MAX_WORKERS);                              // on the caller's thread: no race with a later ioInit()
    for (int t = 0; t < c->ioWorkers; t++)
        for (int k = 0; k < 2; k++)
            if (!c->io[t].done[k]) HIP_CHECK(hipEventCreateWithFlags(&c->io[t].done[k], hipEventDisableTiming));
    u->th = std::thread([=] {
        try { (void)hipSetDevice(c->device); staged_copy(c, (char *)d_dst, src, len, true); }
        catch (const BfqError &e) { u->err = e; u->failed = true; }
        catch (...) { u->ex = std::current_exception(); }
    });
    return u;
}
void bfq_upload_join(BfqAsyncUpload *u)
{
    if (!u) return;
    if (u->th.joinable()) u->th.join();
    const bool failed = u->failed;
    const BfqError err = u->err;
    std::exception_ptr ex = u->ex;
    delete u;
    if (failed) throw err;
    if (ex) std::rethrow_exception(ex);
}

// ---------------------------------------------------------------- host-side line index
// counts[i] = number of '\n' in bytes [i*chunk, (i+1)*chunk) of the text
static void count_range(const uint8_t *h, uint64_t len, uint64_t chunk, uint64_t *counts, uint64_t c0, uint64_t c1)
{
    for (uint64_t ci = c0; ci < c1; ci++) {
        const uint64_t b = ci * chunk, e = std::min(len, b + chunk);
        uint64_t k = 0;
        const uint8_t *p = h + b;
        const uint64_t m = e - b;
        for (uint64_t i = 0; i < m; i++) k += (p[i] == (uint8_t)'\n');   // vectorised by the compiler
        counts[ci] = k;
    }
}
extern "C" int bfq_text_count_lines(const uint8_t *h, uint64_t len, uint64_t chunk, uint64_t *counts, int threads)
{
    if (!chunk || (len && (!h || !counts))) return BFQ_E_ARG;
    const uint64_t nch = (len + chunk - 1) / chunk;
    int T = threads > 0 ? threads : std::max(1, std::min(bfq_cpu_budget(), 16));
    if ((uint64_t)T > nch) T = (int)(nch ? nch : 1);
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(count_range, h, len, chunk, counts, nch * t / T, nch * (t + 1) / T);
    count_range(h, len, chunk, counts, 0, nch / T);
    for (auto &x : th) x.join();
    return BFQ_OK;
}
extern "C" int64_t bfq_text_nth_newline(const uint8_t *h, uint64_t len, uint64_t k)
{
    for (uint64_t i = 0; i < len; i++)
        if (h[i] == (uint8_t)'\n' && k-- == 0) return (int64_t)i;
    return -1;
}

// number of terminator bytes of an eBWT: sizes the outputs of bfq_smooth_invert (N reads, n - N bases)
static void count_byte_range(const uint8_t *h, uint64_t b, uint64_t e, uint8_t v, uint64_t *out)
{
    uint64_t k = 0;
    for (uint64_t i = b; i < e; i++) k += (h[i] == v);
    *out = k;
}
extern "C" int bfq_count_reads(const uint8_t *h_bwt, uint64_t n, int term, uint64_t *N)
{
    if (!N || (n && !h_bwt)) return BFQ_E_ARG;
    int T = n > (64u << 20) ? std::max(1, std::min(bfq_cpu_budget(), 16)) : 1;
    std::vector<uint64_t> part(T, 0);
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(count_byte_range, h_bwt, n * t / T, n * (t + 1) / T, (uint8_t)term, &part[t]);
    count_byte_range(h_bwt, 0, n / T, (uint8_t)term, &part[0]);
    for (auto &x : th) x.join();
    uint64_t k = 0;
    for (int t = 0; t < T; t++) k += part[t];
    *N = k;
    return BFQ_OK;
}
