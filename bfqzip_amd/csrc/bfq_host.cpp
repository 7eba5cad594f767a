// bfq_host.cpp -- host-only pieces of libbfqhip.so (no HIP call in this file):
//   * the per-GPU lease the one-shot tools take (bfq_device_lease): the reference's parallel driver starts n concurrent
//     `BFQzip.py` children, each of which runs `external/gsufsort/gsufsort` and then `src_int_mem/bfq_int`
//     (BFQzip_parallel.py:277-285, BFQzip.py:178-228); the drop-in tools must spread over the node's GPUs by themselves,
//     and never stack several multi-GiB workspaces on one of them;
//   * the phase timeline the tools print with -V / BFQ_TRACE (bfq_phase, bfq_phase_report): bench.py's dropin_wall_s split;
//   * background pre-faulting of an output mapping (bfq_prefault_*): the pages of a 9 GB output file on tmpfs are allocated
//     and zeroed by the kernel at ~6 GB/s; done by helper threads while the GPU works, the final copy runs at memcpy speed.
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "bfq_internal_host.h"

static double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

// ---------------------------------------------------------------- environment, read once
BfqEnv bfq_env_read()
{
    {
        BfqEnv e;
        auto geti = [](const char *k, int dflt) { const char *v = getenv(k); return (v && *v) ? atoi(v) : dflt; };
        auto getu = [](const char *k, unsigned long long dflt) {
            const char *v = getenv(k);
            if (!v || !*v) return dflt;
            char *end = nullptr;
            double x = strtod(v, &end);
            if (end && (*end == 'G' || *end == 'g')) x *= 1073741824.0;
            else if (end && (*end == 'M' || *end == 'm')) x *= 1048576.0;
            else if (end && (*end == 'K' || *end == 'k')) x *= 1024.0;
            return (unsigned long long)x;
        };
        e.trace = getenv("BFQ_TRACE") != nullptr;
        e.piles = getenv("BFQ_PILES") ? (atoi(getenv("BFQ_PILES")) == 2 ? 2 : atoi(getenv("BFQ_PILES")) ? 1 : -1) : 0;
        e.pilesSplit = getenv("BFQ_PILES_SPLIT") != nullptr;
        e.noOverlap = getenv("BFQ_NO_OVERLAP") != nullptr;
        e.noLengthGuess = getenv("BFQ_NO_LENGTH_GUESS") != nullptr;
        e.posMode = geti("BFQ_POSMODE", 0) != 0;
        e.invertThreads = geti("BFQ_INVERT_THREADS", 0);
        e.ioThreads = geti("BFQ_IO_THREADS", 0);
        e.prefaultThreads = geti("BFQ_PREFAULT_THREADS", -1);
        e.hugeCap = getu("BFQ_HUGE_CAP", 0);
        e.wsCap = getu("BFQ_WS_CAP", 0);
        e.device = getenv("BFQ_DEVICE") ? atoi(getenv("BFQ_DEVICE")) : -1;
        e.fakeDevices = geti("BFQ_FAKE_DEVICES", 0);
        e.lease = geti("BFQ_LEASE", 1) != 0;
        if (const char *d = getenv("BFQ_LEASE_DIR")) e.leaseDir = d;
        e.invertNt = geti("BFQ_INVERT_NT", 1);
        e.noOutmap = getenv("BFQ_NO_OUTMAP") != nullptr;
        e.wsContig = geti("BFQ_WS_CONTIG", 0) != 0;
        e.compact = geti("BFQ_COMPACT", 0) != 0;
        e.prefaultPause = geti("BFQ_PREFAULT_PAUSE", 0) != 0;
        e.compactWin = getu("BFQ_COMPACT_WIN", 0);
        e.compactRing = getu("BFQ_COMPACT_RING", 0);
        e.dnaStatic = geti("BFQ_DNA_STATIC", 0);
        e.keyFusion = geti("BFQ_KEY_FUSION", 0) != 0;
        e.dnacK = geti("BFQ_DNAC_K", 0); e.dnacH = geti("BFQ_DNAC_H", 0); e.dnacW = geti("BFQ_DNAC_W", 0); e.dnacSkip = geti("BFQ_DNAC_TSKIP", -1);
        e.wsVmmMib = geti("BFQ_WS_VMM", 0);
        e.rsPerm = geti("BFQ_RS_PERM", 0) != 0;
        e.abPad = getu("BFQ_AB_PAD", 0);
        e.abSwap = geti("BFQ_AB_SWAP", 0) != 0;
        if (const char *o = getenv("BFQ_AB_ORDER")) {
            bool seen[4] = {false, false, false, false};
            bool ok = strlen(o) == 4;
            for (int k = 0; ok && k < 4; k++) { ok = o[k] >= '0' && o[k] <= '3' && !seen[o[k] - '0']; if (ok) seen[o[k] - '0'] = true; }
            if (ok) memcpy(e.abOrder, o, 4);
        }
        return e;
    }
}
const BfqEnv &bfq_env()
{
    static const BfqEnv E = bfq_env_read();
    return E;
}

// CPUs this process may keep busy: the cgroup's quota (cpu.max) when there is one -- a GPU box hands a 256-thread host out in
// shares of 16 CPUs, and a process that runs more threads than its quota is throttled as a whole, the thread that feeds
// the GPU included -- else the affinity mask / hardware concurrency.
int bfq_cpu_budget()
{
    static const int budget = [] {
        int n = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0 && (n <= 0 || k < n)) n = k; }
        for (const char *path : {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"}) {
            FILE *f = fopen(path, "r");
            if (!f) continue;
            char a[64] = {0}, b[64] = {0};
            const int got = fscanf(f, "%63s %63s", a, b);
            fclose(f);
            if (got >= 1 && strcmp(a, "max") != 0) {
                double quota = atof(a), period = got >= 2 ? atof(b) : 100000.0;
                if (got < 2) {                                   // cgroup v1: the period lives in its own file
                    FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
                    if (g) { if (fscanf(g, "%lf", &period) != 1) period = 100000.0; fclose(g); }
                }
                if (quota > 0 && period > 0) { const int k = (int)(quota / period + 0.5); if (k >= 1 && (n <= 0 || k < n)) n = k; }
            }
            break;
        }
        return n > 0 ? n : 1;
    }();
    return budget;
}

// ---------------------------------------------------------------- device lease
// One lock file per device slot; a slot is held by flock(LOCK_EX) on a descriptor that stays open until
// bfq_device_release() or the end of the process (the kernel drops the lock with the descriptor, also when the process is
// killed).  Free slots are tried in order, so eight concurrent tools on an 8-GPU node end up on eight GPUs; when every
// slot is taken the caller polls until one is released: tools on one GPU run one after the other.
namespace {
struct Held { int slot; int fd; std::string path; };
std::mutex g_leaseMu;
std::vector<Held> g_held;

std::string lease_dir()
{
    const BfqEnv &E = bfq_env();
    if (!E.leaseDir.empty()) return E.leaseDir;
    if (access("/dev/shm", W_OK | X_OK) == 0) return "/dev/shm";
    return "/tmp";
}
}   // namespace

extern "C" int bfq_device_lease(int n_slots, const char *const *slot_ids, int only_slot, char *path_out, int path_cap,
                                double *waited_s)
{
    if (n_slots < 1) return BFQ_E_ARG;
    if (only_slot >= n_slots) return BFQ_E_ARG;
    const std::string dir = lease_dir();
    std::vector<int> fds(n_slots, -1);
    std::vector<std::string> paths(n_slots);
    const mode_t um = umask(0);                                  // the lock files are shared by every user of the node
    for (int k = 0; k < n_slots; k++) {
        if (only_slot >= 0 && k != only_slot) continue;
        std::string id = (slot_ids && slot_ids[k]) ? slot_ids[k] : ("slot" + std::to_string(k));
        for (auto &ch : id) if (!((ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || ch == '.' || ch == '-')) ch = '_';
        paths[k] = dir + "/bfqzip_amd." + id + ".lock";
        fds[k] = open(paths[k].c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
    }
    umask(um);
    int got = -1;
    const double t0 = now_s();
    unsigned spin = 0;
    bool any = false;
    for (int k = 0; k < n_slots; k++) any = any || fds[k] >= 0;
    if (!any) return BFQ_E_IO;
    while (got < 0) {
        for (int k = 0; k < n_slots && got < 0; k++) {
            if (fds[k] < 0) continue;
            if (flock(fds[k], LOCK_EX | LOCK_NB) == 0) got = k;
            else if (errno != EWOULDBLOCK && errno != EINTR) { close(fds[k]); fds[k] = -1; }
        }
        if (got >= 0) break;
        any = false;
        for (int k = 0; k < n_slots; k++) any = any || fds[k] >= 0;
        if (!any) return BFQ_E_IO;
        const unsigned us = spin < 20 ? 2000 : spin < 100 ? 10000 : 50000;   // 2 ms, then 10 ms, then 50 ms between rounds
        usleep(us);
        spin++;
    }
    for (int k = 0; k < n_slots; k++)
        if (k != got && fds[k] >= 0) close(fds[k]);
    if (waited_s) *waited_s = now_s() - t0;
    if (path_out && path_cap > 0) { strncpy(path_out, paths[got].c_str(), path_cap - 1); path_out[path_cap - 1] = 0; }
    {   // who holds it, for people looking at the directory (best effort)
        char b[64];
        int l = snprintf(b, sizeof b, "%ld\n", (long)getpid());
        if (ftruncate(fds[got], 0) == 0) { ssize_t w = pwrite(fds[got], b, (size_t)l, 0); (void)w; }
    }
    std::lock_guard<std::mutex> g(g_leaseMu);
    g_held.push_back(Held{got, fds[got], paths[got]});
    return got;
}

extern "C" int bfq_device_release(int slot)
{
    std::lock_guard<std::mutex> g(g_leaseMu);
    for (size_t i = 0; i < g_held.size(); i++)
        if (g_held[i].slot == slot) {
            flock(g_held[i].fd, LOCK_UN);
            close(g_held[i].fd);
            g_held.erase(g_held.begin() + (long)i);
            return BFQ_OK;
        }
    return BFQ_E_ARG;
}

// ---------------------------------------------------------------- phase timeline
namespace {
struct Phase { std::string name; double secs; };
std::mutex g_phMu;
std::vector<Phase> g_phases;
double g_phT0 = 0, g_phLast = 0;
int g_phCur = -1;
bool g_phOn = false;
double g_startAge = -1;                                          // seconds between exec and the first mark

// seconds since this process was started (exec): /proc/self/stat field 22 against /proc/uptime, 10 ms resolution
double process_age()
{
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return -1;
    char buf[2048];
    size_t r = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[r] = 0;
    const char *p = strrchr(buf, ')');                           // the command name may contain spaces
    if (!p) return -1;
    p++;
    unsigned long long start = 0;
    int field = 3;                                               // the next token is field 3 (state); starttime is field 22
    while (*p && field < 22) { while (*p == ' ') p++; while (*p && *p != ' ') p++; field++; }
    if (sscanf(p, " %llu", &start) != 1) return -1;
    double up = 0;
    f = fopen("/proc/uptime", "r");
    if (!f) return -1;
    int ok = fscanf(f, "%lf", &up);
    fclose(f);
    if (ok != 1) return -1;
    const long hz = sysconf(_SC_CLK_TCK);
    return up - (double)start / (double)(hz > 0 ? hz : 100);
}
}   // namespace

extern "C" void bfq_phase_enable(int on)
{
    std::lock_guard<std::mutex> g(g_phMu);
    g_phOn = on != 0;
}
extern "C" void bfq_phase(const char *name)
{
    std::lock_guard<std::mutex> g(g_phMu);                       // always recorded (a clock read); printed only when asked for
    const double t = now_s();
    if (g_phCur < 0 && g_phases.empty()) { g_phT0 = t; g_phLast = t; g_startAge = process_age(); }
    if (g_phCur >= 0) g_phases[(size_t)g_phCur].secs += t - g_phLast;
    g_phLast = t;
    g_phCur = -1;
    if (!name) return;
    for (size_t i = 0; i < g_phases.size(); i++)
        if (g_phases[i].name == name) g_phCur = (int)i;
    if (g_phCur < 0) { g_phases.push_back(Phase{name, 0}); g_phCur = (int)g_phases.size() - 1; }
}
// one line: [bfq phases] {"tool": ..., "exec_to_main": s, "<phase>": s, ..., "total": s}
extern "C" void bfq_phase_report(const char *tool)
{
    bfq_phase(nullptr);
    std::lock_guard<std::mutex> g(g_phMu);
    if (g_phases.empty() || !(g_phOn || bfq_env().trace)) return;
    std::string s = "[bfq phases] {\"tool\": \"";
    s += tool ? tool : "";
    s += "\"";
    char b[128];
    if (g_startAge >= 0) { snprintf(b, sizeof b, ", \"exec_to_main\": %.3f", g_startAge); s += b; }
    for (auto &p : g_phases) { snprintf(b, sizeof b, ", \"%s\": %.3f", p.name.c_str(), p.secs); s += b; }
    snprintf(b, sizeof b, ", \"total\": %.3f}", (g_startAge >= 0 ? g_startAge : 0) + (g_phLast - g_phT0));
    s += b;
    fprintf(stderr, "%s\n", s.c_str());
    fflush(stderr);
}

// ---------------------------------------------------------------- output mapping + background pre-fault
// Measured on the GPU boxes (profiles/microbench/tmpfs_fill.cpp, 8.6 GB on /dev/shm): MADV_POPULATE_WRITE of a new file runs
// at 6.5 GB/s with 4 threads and SLOWER with more (4.0 GB/s with 8, 3.7 with 16: they contend inside the kernel); a memcpy
// that takes the faults itself is no faster.  fallocate() allocates and zeroes the same pages at 19 GB/s from ONE thread
// (no page-table work), MADV_POPULATE_WRITE of pages that exist already then maps them at 43 GB/s, and a memcpy into
// populated pages runs at 100+ GB/s.  So: few helper threads go through the file slice by slice -- fallocate, then
// populate -- and whoever wants to copy into a slice waits until it is done instead of faulting beside them (a writer
// prepares a slice itself only when no helper will).
struct bfq_outmap {
    int fd = -1;
    char *map = nullptr;
    uint64_t mapLen = 0;
    std::vector<std::thread> th;
    std::atomic<uint64_t> next{0};                               // next slice index a helper takes
    std::atomic<bool> stop{false};
    std::atomic<int> alive{0};
    std::atomic<uint64_t> preEnd{0};
    std::atomic<uint64_t> done{0};
    uint64_t nslices = 0;
    std::atomic<bool> noFallocate{false};
    bool holdForHip = false;                                     // opened before the process had a context: see prefault_worker
    std::atomic<uint64_t> fallocDone{0};                         // bytes from the start of the file whose pages exist
    std::atomic<bool> fallocRunning{false};
    std::thread fallocThread;
    std::atomic<bool> failed{false};                             // pages could not be had (ENOSPC ...): writers use pwrite and report it
    double tOpen = 0, tHelpersDone = 0;
    std::atomic<uint8_t> *state = nullptr;                       // per slice: 0 untouched, 1 being populated, 2 populated
};
static const uint64_t PF_SLICE = 32ull << 20;

// at most this many threads of the process populate at a time, whatever the number of mappings (more are slower in total)
namespace {
std::mutex g_popMu;
std::condition_variable g_popCv;
int g_popBusy = 0;
struct PopSlot {
    PopSlot()
    {
        int lim = bfq_env().prefaultThreads;
        if (lim <= 0) lim = 2;                                   // populate of present pages: 2 threads map 24 GB/s
        std::unique_lock<std::mutex> lk(g_popMu);
        g_popCv.wait(lk, [&] { return g_popBusy < lim; });
        g_popBusy++;
    }
    ~PopSlot() { std::lock_guard<std::mutex> g(g_popMu); g_popBusy--; g_popCv.notify_one(); }
};
}   // namespace
// staged uploads in progress (bfq_io.hip): the helpers stand back while a file is being read into the staging buffers --
// reading 9.5 GB of page cache and allocating 9 GB of it at the same time is slower than one after the other (both work the
// same LRU lists; BFQ_PREFAULT_PAUSE=0 lets them overlap)
std::atomic<int> g_bfqUploadsRunning{0};
static void populate_slice(bfq_outmap *m, uint64_t idx)
{
    const uint64_t b = idx * PF_SLICE, e = std::min<uint64_t>(b + PF_SLICE, m->mapLen);
    if (m->failed.load()) return;
    // the pages are allocated by ONE thread per file that runs ahead (falloc_worker: concurrent fallocate calls on one file are
    // slower than one, 7 against 19 GB/s); a slice it will not reach (beyond the pre-fault range, or no such thread) is
    // allocated here
    while (!m->noFallocate.load() && !m->failed.load() && m->fallocDone.load() < e) {
        if (m->fallocRunning.load() && e <= m->preEnd.load()) { usleep(200); continue; }
        if (fallocate(m->fd, 0, (off_t)b, (off_t)(e - b)) != 0) {
            if (errno == ENOSPC || errno == EDQUOT || errno == EFBIG) { m->failed = true; return; }
            m->noFallocate = true;
        }
        break;
    }
    if (m->failed.load()) return;
    PopSlot slot;
#ifdef MADV_POPULATE_WRITE
    if (madvise(m->map + b, (size_t)(e - b), MADV_POPULATE_WRITE) != 0) {
        if (errno == ENOMEM || errno == EFAULT || errno == ENOSPC) { m->failed = true; return; }   // (EFAULT: the fault would have been a SIGBUS)
        if (errno != EINVAL) { m->failed = true; return; }
        // EINVAL: a kernel without MADV_POPULATE_WRITE: the copy takes the faults itself
    }
#endif
    m->done += e - b;
}
// allocates (and zeroes) the file's pages from the front, 128 MiB per call
extern std::atomic<bool> g_bfqHipStarted;
static void hold_for_hip(bfq_outmap *m);
static void falloc_worker(bfq_outmap *m)
{
    hold_for_hip(m);
    const uint64_t STEP = 128ull << 20;
    for (uint64_t b = 0;;) {
        if (m->stop.load(std::memory_order_relaxed) || m->failed.load()) break;
        if (bfq_env().prefaultPause && g_bfqUploadsRunning.load(std::memory_order_relaxed) > 0) { usleep(500); continue; }
        const uint64_t pe = m->preEnd.load();
        if (b >= pe) break;
        const uint64_t e = std::min<uint64_t>(b + STEP, std::min<uint64_t>(pe + PF_SLICE, m->mapLen));
        if (fallocate(m->fd, 0, (off_t)b, (off_t)(e - b)) != 0) {
            // no room (or a quota): touching the mapping would end in SIGBUS -- from here on this file is written with pwrite,
            // which reports the error.  Anything else: a file system without fallocate, the populate does the allocation.
            if (errno == ENOSPC || errno == EDQUOT || errno == EFBIG) m->failed = true; else m->noFallocate = true;
            break;
        }
        m->fallocDone = e;
        b = e;
    }
    m->fallocRunning = false;
}
// MADV_POPULATE_WRITE holds the address space's lock for reading while it maps pages; the HIP runtime's start is thousands of
// mmap / ioctl calls that take it for writing: with the populate helpers of two output files running, hipGetDeviceCount took
// 1.0 s instead of 0.15.  The helpers that map pages therefore wait until a context exists (bfq_create says so) -- the
// thread that allocates the pages (fallocate: the file, not the address space) starts at once.  At most 3 s: a caller that
// never creates a context still gets its pages.
std::atomic<bool> g_bfqHipStarted{false};
static void hold_for_hip(bfq_outmap *m)
{
    if (m->holdForHip)
        for (double t0 = now_s(); !g_bfqHipStarted.load(std::memory_order_relaxed) && !m->stop.load(std::memory_order_relaxed) && now_s() - t0 < 3.0;) usleep(1000);
}
static void prefault_worker(bfq_outmap *m)
{
    hold_for_hip(m);
    for (;;) {
        if (m->stop.load(std::memory_order_relaxed)) break;
        const uint64_t idx = m->next.fetch_add(1);
        if (idx >= m->nslices || idx * PF_SLICE >= m->preEnd.load()) { m->next.fetch_sub(1); break; }
        uint8_t z = 0;
        if (m->state[idx].compare_exchange_strong(z, 1)) { populate_slice(m, idx); m->state[idx].store(2); }
    }
    if (--m->alive == 0) m->tHelpersDone = now_s();
}
// the bytes [off, off + len) of the mapping are about to be written: returns once their pages are populated
// false: the pages cannot be had -- write the bytes with pwrite(bfq_outmap_fd(m), ...) instead, it will say why
bool bfq_outmap_ensure(bfq_outmap *m, uint64_t off, uint64_t len)
{
    if (!m || !len) return true;
    if (m->failed.load()) return false;
    const uint64_t s0 = off / PF_SLICE, s1 = std::min<uint64_t>((off + len - 1) / PF_SLICE, m->nslices - 1);
    for (uint64_t idx = s0; idx <= s1; idx++) {
        for (;;) {
            const uint8_t st = m->state[idx].load();
            if (st == 2) break;
            // nobody is going to do it (beyond the pre-fault range, or the helpers have gone): do it here
            if (st == 0 && (m->alive.load() == 0 || idx * PF_SLICE >= m->preEnd.load() || idx < m->next.load())) {
                uint8_t z = 0;
                if (m->state[idx].compare_exchange_strong(z, 1)) { populate_slice(m, idx); m->state[idx].store(2); break; }
            }
            if (m->failed.load()) return false;
            usleep(200);
        }
    }
    return !m->failed.load();
}
int bfq_outmap_fd(bfq_outmap *m) { return m ? m->fd : -1; }

// Sizes fd to map_len bytes, maps it shared and starts helper threads that fault in [0, prefault_len).  Returns nullptr
// when the descriptor cannot be mapped (a pipe, /dev/null ...): the caller then writes with pwrite.
bfq_outmap *bfq_outmap_open(int fd, uint64_t map_len, uint64_t prefault_len)
{
    if (fd < 0 || !map_len || bfq_env().noOutmap) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return nullptr;
    if (ftruncate(fd, (off_t)map_len) != 0) return nullptr;
    void *p = mmap(nullptr, (size_t)map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (p == MAP_FAILED) { if (ftruncate(fd, 0) != 0) {} return nullptr; }
    bfq_outmap *m = new bfq_outmap();
    m->fd = fd; m->map = (char *)p; m->mapLen = map_len;
    m->nslices = (map_len + PF_SLICE - 1) / PF_SLICE;
    m->state = new std::atomic<uint8_t>[m->nslices];
    for (uint64_t i = 0; i < m->nslices; i++) m->state[i].store(0);
    m->preEnd = prefault_len < map_len ? prefault_len : map_len;
    const uint64_t pe0 = m->preEnd.load();
    int T = bfq_env().prefaultThreads;
    if (T < 0) {
        T = bfq_cpu_budget() >= 8 ? 2 : 1;                       // (+ the thread that allocates ahead of them)
    }
    if (pe0 < (64ull << 20)) T = 0;                              // small outputs: not worth a thread
    m->alive = T;
    m->tOpen = now_s();
    m->holdForHip = !g_bfqHipStarted.load();
    if (T) { m->fallocRunning = true; m->fallocThread = std::thread(falloc_worker, m); }
    for (int t = 0; t < T; t++) m->th.emplace_back(prefault_worker, m);
    return m;
}
char *bfq_outmap_ptr(bfq_outmap *m) { return m ? m->map : nullptr; }
uint64_t bfq_outmap_len(bfq_outmap *m) { return m ? m->mapLen : 0; }
// more of the file becomes certain (e.g. once the record count is known): extend the pre-fault range
void bfq_outmap_extend(bfq_outmap *m, uint64_t prefault_len)
{
    if (!m) return;
    if (prefault_len > m->mapLen) prefault_len = m->mapLen;
    if (prefault_len <= m->preEnd.load()) return;
    // the helpers read preEnd on every slice: raising it keeps those still alive going; finished ones are not restarted
    // (whoever writes there populates what they did not reach)
    m->preEnd = prefault_len;
}
// stops the helpers, unmaps, cuts the file to its final length; false if that failed
bool bfq_outmap_close(bfq_outmap *m, uint64_t final_len)
{
    if (!m) return true;
    m->stop = true;
    if (m->fallocThread.joinable()) m->fallocThread.join();
    for (auto &t : m->th) if (t.joinable()) t.join();
    if (bfq_env().trace)
        fprintf(stderr, "[bfq io] output mapping %.2f GB: %.2f GB prepared (fallocate + populate), %zu helper thread(s) done %.3f s after the file was opened (closed after %.3f s)\n",
                final_len / 1e9, (double)m->done.load() / 1e9, m->th.size(), m->tHelpersDone > 0 ? m->tHelpersDone - m->tOpen : -1.0, now_s() - m->tOpen);
    bool ok = munmap(m->map, (size_t)m->mapLen) == 0;
    if (final_len != m->mapLen) ok = (ftruncate(m->fd, (off_t)final_len) == 0) && ok;
    delete[] m->state;
    delete m;
    return ok;
}

// ---------------------------------------------------------------- what a tool can do before the GPU is even initialised
// eBWT rows of a FASTQ file, estimated from the complete records among its first bytes (0.98 of it: the estimate sizes the
// background pre-fault of the outputs before the file has been parsed; a wrong guess costs time, never correctness)
extern "C" uint64_t bfq_fastq_rows_estimate(int fd, uint64_t len)
{
    const size_t want = (size_t)(len < (1u << 20) ? len : (1u << 20));
    if (!want || fd < 0) return 0;
    std::vector<uint8_t> b(want);
    double rpb = 0.45;
    if (pread(fd, b.data(), want, 0) == (ssize_t)want) {
        uint64_t rows = 0, used = 0;
        size_t pos = 0;
        for (;;) {
            size_t e[4], p = pos;
            int k = 0;
            for (; k < 4; k++) {
                const void *q = p < want ? memchr(b.data() + p, '\n', want - p) : nullptr;
                if (!q) break;
                e[k] = (size_t)((const uint8_t *)q - b.data());
                p = e[k] + 1;
            }
            if (k < 4) break;
            size_t L = e[1] - (e[0] + 1);
            if (L && b[e[1] - 1] == '\r') L--;
            rows += L + 1;
            used = p;
            pos = p;
        }
        if (used) rpb = (double)rows / (double)used;
    }
    if (rpb > 0.5) rpb = 0.5;
    return (uint64_t)(rpb * (double)len * 0.98);
}

// A tool opens its outputs first thing and registers them here; the entry point that later gets the same descriptor finds
// the mapping already there and (mostly) faulted in.
namespace {
std::mutex g_regMu;
std::vector<bfq_outmap *> g_reg;
}
extern "C" int bfq_output_prefault(int fd, uint64_t map_len, uint64_t prefault_len)
{
    bfq_outmap *m = bfq_outmap_open(fd, map_len, prefault_len);
    if (!m) return BFQ_E_IO;
    std::lock_guard<std::mutex> g(g_regMu);
    g_reg.push_back(m);
    return BFQ_OK;
}
bfq_outmap *bfq_outmap_take(int fd, uint64_t min_len)
{
    std::lock_guard<std::mutex> g(g_regMu);
    for (size_t i = 0; i < g_reg.size(); i++)
        if (g_reg[i]->fd == fd) {
            bfq_outmap *m = g_reg[i];
            g_reg.erase(g_reg.begin() + (long)i);
            if (m->mapLen >= min_len) return m;
            bfq_outmap_close(m, 0);                              // too small for what is coming: start over
            return nullptr;
        }
    return nullptr;
}

// ---------------------------------------------------------------- a host buffer into a file at an offset, fast
// What the multi-GPU driver does with every block's outputs (BFQzip_parallel.py:137-179 merges with `cat` and Python line
// loops; bfqzip_amd/parallel.py writes every rank's bytes at their final offsets of the shared output files).  A Python
// os.pwrite moves 9.5 GB to tmpfs in 1.6 s (one thread; several writers of one file serialise on its inode lock); here the
// range is fallocate()d (extends the file if needed -- it never shrinks it, so ranks writing different ranges of one file do
// not disturb each other), mapped, and a few threads populate and fill it slice by slice.
extern "C" int bfq_file_put(int fd, uint64_t off, const void *src, uint64_t len, int threads)
{
    if (fd < 0 || (len && !src)) return BFQ_E_ARG;
    if (!len) return BFQ_OK;
    auto pwrite_all = [&](uint64_t o, const char *p, uint64_t n) {
        while (n) {
            const ssize_t w = pwrite(fd, p, (size_t)(n > (1ull << 30) ? (1ull << 30) : n), (off_t)o);
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) return false;
            o += (uint64_t)w; p += w; n -= (uint64_t)w;
        }
        return true;
    };
    const long pg = sysconf(_SC_PAGESIZE);
    const uint64_t a0 = off / (uint64_t)pg * (uint64_t)pg;
    char *map = nullptr;
    if (!bfq_env().noOutmap && fallocate(fd, 0, (off_t)off, (off_t)len) == 0) {
        void *m = mmap(nullptr, (size_t)(off + len - a0), PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)a0);
        if (m != MAP_FAILED) map = (char *)m;
    }
    if (!map) return pwrite_all(off, (const char *)src, len) ? BFQ_OK : BFQ_E_IO;
    int T = threads > 0 ? threads : (bfq_cpu_budget() >= 16 ? 6 : bfq_cpu_budget() >= 8 ? 4 : 2);
    const uint64_t S = 32ull << 20, ns = (len + S - 1) / S;
    if ((uint64_t)T > ns) T = (int)ns;
    std::atomic<uint64_t> next{0};
    char *dst = map + (off - a0);
    auto work = [&] {
        for (;;) {
            const uint64_t i = next.fetch_add(1);
            if (i >= ns) return;
            const uint64_t b = i * S, e = b + S < len ? b + S : len;
#ifdef MADV_POPULATE_WRITE
            {   // page-aligned superset of the slice (the first one starts inside a page when off is not aligned)
                const uint64_t pb = (uint64_t)(dst + b - map) / (uint64_t)pg * (uint64_t)pg, pe = (uint64_t)(dst + e - map);
                (void)madvise(map + pb, (size_t)(pe - pb), MADV_POPULATE_WRITE);
            }
#endif
            memcpy(dst + b, (const char *)src + b, (size_t)(e - b));
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    return munmap(map, (size_t)(off + len - a0)) == 0 ? BFQ_OK : BFQ_E_IO;
}

// The same for a caller that wants to fill the range itself (the library's transfers then copy straight into the file's
// pages instead of into a buffer that is written out afterwards): the byte range [off, off + len) of the file, allocated
// (fallocate), mapped and populated by a few threads; nullptr when the file cannot be mapped.  bfq_file_unmap() with the
// same off / len gives it back (the pages stay in the page cache: the data is written).
extern "C" void *bfq_file_map(int fd, uint64_t off, uint64_t len, int threads)
{
    if (fd < 0 || !len || bfq_env().noOutmap) return nullptr;
    const long pg = sysconf(_SC_PAGESIZE);
    const uint64_t a0 = off / (uint64_t)pg * (uint64_t)pg;
    if (fallocate(fd, 0, (off_t)off, (off_t)len) != 0) return nullptr;
    void *m = mmap(nullptr, (size_t)(off + len - a0), PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)a0);
    if (m == MAP_FAILED) return nullptr;
#ifdef MADV_POPULATE_WRITE
    int T = threads > 0 ? threads : (bfq_cpu_budget() >= 8 ? 4 : 2);
    const uint64_t S = 64ull << 20, span = off + len - a0, ns = (span + S - 1) / S;
    if ((uint64_t)T > ns) T = (int)ns;
    std::atomic<uint64_t> next{0};
    auto work = [&] {
        for (;;) {
            const uint64_t i = next.fetch_add(1);
            if (i >= ns) return;
            const uint64_t b = i * S, e = b + S < span ? b + S : span;
            (void)madvise((char *)m + b, (size_t)(e - b), MADV_POPULATE_WRITE);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
#else
    (void)threads;
#endif
    return (char *)m + (off - a0);
}
extern "C" int bfq_file_unmap(void *p, uint64_t off, uint64_t len)
{
    if (!p) return BFQ_E_ARG;
    const long pg = sysconf(_SC_PAGESIZE);
    const uint64_t a0 = off / (uint64_t)pg * (uint64_t)pg;
    return munmap((char *)p - (off - a0), (size_t)(off + len - a0)) == 0 ? BFQ_OK : BFQ_E_IO;
}
