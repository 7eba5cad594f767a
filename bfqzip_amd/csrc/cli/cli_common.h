// cli_common.h -- file helpers shared by the drop-in front-ends (host only).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../../include/bfqzip_hip.h"

static inline bool read_file(const std::string &path, std::vector<uint8_t> &buf)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = sz > 0 ? fread(buf.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    return got == buf.size();
}
static inline bool write_file(const std::string &path, const void *p, size_t n)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    size_t w = n ? fwrite(p, 1, n, f) : 0;
    fclose(f);
    return w == n;
}
static inline bool file_exists(const std::string &p)
{
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    fclose(f);
    return true;
}

// ---- files as descriptors: the front-ends hand open files to the library, whose pinned staging pipeline moves the
// bytes between them and the GPU (bfq_io.hip); nothing is mapped, copied or zero-filled on the way.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

// The GPU this process uses: the first one nobody else holds (per-GPU lease files, include/bfqzip_hip.h), so that the n
// concurrent children of an unchanged BFQzip_parallel.py (:277-285) spread over the node's GPUs; $BFQ_DEVICE pins it.
static inline bfq_ctx *create_on_free_gpu(const char *tool, const bfq_params *P)
{
    bfq_phase("lease");
    char info[320] = {0};
    const int dev = bfq_pick_device(info, sizeof info);
    if (dev < 0) { fprintf(stderr, "%s: no GPU to run on (bfq_pick_device: %d)\n", tool, dev); return nullptr; }
    if (getenv("BFQ_TRACE")) fprintf(stderr, "[bfq lease] %s pid %ld: %s\n", tool, (long)getpid(), info);
    bfq_phase("hip_init");
    bfq_ctx *c = bfq_create(dev, P);
    if (!c) fprintf(stderr, "%s: %s\n", tool, bfq_create_error());
    return c;
}

// BFQ_TRACE: the kernels' accumulated HIP-event times of this process
static inline void trace_kernel_times(bfq_ctx *c, const char *tool)
{
    if (!getenv("BFQ_TRACE")) return;
    std::string s = std::string("[bfq kernels] ") + tool + ":";
    for (int i = 0; i < bfq_prof_count(c); i++) {
        char name[64];
        double ms = 0, bytes = 0;
        uint64_t launches = 0;
        if (bfq_prof_get(c, i, name, sizeof name, &ms, &launches, &bytes) == 0 && launches) {
            char b[128];
            snprintf(b, sizeof b, " %s %.1f ms/%llu", name, ms, (unsigned long long)launches);
            s += b;
        }
    }
    fprintf(stderr, "%s\n", s.c_str());
}

struct InFile {
    int fd = -1;
    uint64_t size = 0;
    bool open(const std::string &path)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        size = (uint64_t)st.st_size;
        return true;
    }
    ~InFile() { if (fd >= 0) ::close(fd); }
};
struct OutFile {
    int fd = -1;
    bool open(const std::string &path)
    {
        fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);   // O_RDWR: the library maps the file to write it
        return fd >= 0;
    }
    bool close()
    {
        bool ok = fd < 0 || ::close(fd) == 0;
        fd = -1;
        return ok;
    }
    ~OutFile() { if (fd >= 0) ::close(fd); }
};
