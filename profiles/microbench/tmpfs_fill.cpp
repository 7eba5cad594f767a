// tmpfs_fill.cpp -- how fast can a process fill a NEW multi-GB file on /dev/shm?  (host only; the drop-in tools write 9 GB
// outputs: BFQzip.py:184,215-222 at 30 M x 150 bp.)     g++ -O2 -o tmpfs_fill tmpfs_fill.cpp -lpthread
//   tmpfs_fill <threads> <MiB> <mode> [files]
//   mode 0  madvise(MADV_POPULATE_WRITE) on a shared mapping   (page allocation + zeroing + mapping, by page faults)
//        1  memcpy into the untouched mapping                  (the same faults, taken by the copy itself)
//        2  pwrite                                             (no zeroing, but one writer per file at a time: inode lock)
//        3  fallocate                                          (allocation + zeroing in the kernel, no mapping)
//        4  fallocate, then memcpy into the mapping            (minor write faults only)           -- both phases timed
//        5  fallocate, then pwrite
//        6  MADV_POPULATE_WRITE, then memcpy                   (what libbfqhip.so did first)       -- both phases timed
//        7  fallocate (ONE thread), then MADV_POPULATE_WRITE of the now present pages, then memcpy  -- three phases timed
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include <atomic>
#include <algorithm>
#include <time.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    if (argc < 4) return 1;
    const int T = atoi(argv[1]);
    const size_t len = (size_t)atoll(argv[2]) << 20;
    const int mode = atoi(argv[3]), nfiles = argc > 4 ? atoi(argv[4]) : 1;
    std::vector<char *> maps;
    std::vector<int> fds;
    const size_t per = len / nfiles;
    for (int f = 0; f < nfiles; f++) {
        char p[64];
        snprintf(p, 64, "/dev/shm/tmpfs_fill_%d", f);
        int fd = open(p, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0 || ftruncate(fd, per) != 0) return 2;
        maps.push_back((char *)mmap(0, per, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
        fds.push_back(fd);
    }
    const size_t S = 32u << 20;
    auto pass = [&](int what) {
        std::atomic<size_t> next{0};
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&] {
                char *buf = (char *)malloc(S);
                memset(buf, 1, S);
                for (;;) {
                    const size_t b = next.fetch_add(S);
                    if (b >= len) break;
                    const int f = (int)(b / per);
                    const size_t o = b % per, e = std::min(S, per - o);
                    if (what == 0) madvise(maps[f] + o, e, MADV_POPULATE_WRITE);
                    else if (what == 1) memcpy(maps[f] + o, buf, e);
                    else if (what == 2) { if (pwrite(fds[f], buf, e, o) < 0) abort(); }
                    else if (what == 3) { if (fallocate(fds[f], 0, o, e) != 0) abort(); }
                }
                free(buf);
            });
        for (auto &x : th) x.join();
        return len / 1e9 / (now() - t0);
    };
    printf("threads %d, %d file(s), %.1f GB, mode %d:", T, nfiles, len / 1e9, mode);
    if (mode <= 3) printf(" %.2f GB/s\n", pass(mode));
    else if (mode == 4) { double a = pass(3), b = pass(1); printf(" fallocate %.2f GB/s, then memcpy %.2f GB/s\n", a, b); }
    else if (mode == 5) { double a = pass(3), b = pass(2); printf(" fallocate %.2f GB/s, then pwrite %.2f GB/s\n", a, b); }
    else if (mode == 7) {
        const double t0 = now();
        for (int f = 0; f < nfiles; f++) if (fallocate(fds[f], 0, 0, per) != 0) abort();
        const double a = len / 1e9 / (now() - t0);
        double b = pass(0), c = pass(1);
        printf(" fallocate (1 thread) %.2f GB/s, then populate %.2f GB/s, then memcpy %.2f GB/s\n", a, b, c);
    }
    else if (mode == 6) { double a = pass(0), b = pass(1); printf(" populate %.2f GB/s, then memcpy %.2f GB/s\n", a, b); }
    for (int f = 0; f < nfiles; f++) { char p[64]; snprintf(p, 64, "/dev/shm/tmpfs_fill_%d", f); unlink(p); }
    return 0;
}
