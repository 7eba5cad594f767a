// cli_common.h -- file helpers shared by the drop-in front-ends (host only).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <memory>
#include <new>
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

// ---- files as memory: the front-ends hand mapped files to the library, whose pinned staging pipeline moves
// the bytes (bfq_io.hip); nothing is copied or zero-filled on the way.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

struct MappedInput {                         // read-only view of a whole file
    const uint8_t *data = nullptr;
    size_t size = 0;
    bool open(const std::string &path)
    {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); return false; }
        size = (size_t)st.st_size;
        if (size) {
            void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p == MAP_FAILED) {                                   // not mappable (pipe, odd file system): read it
                own.resize(size);
                size_t got = 0;
                while (got < size) { ssize_t r = ::read(fd, own.data() + got, size - got); if (r <= 0) break; got += (size_t)r; }
                ::close(fd);
                if (got != size) return false;
                data = own.data();
                return true;
            }
            (void)madvise(p, size, MADV_SEQUENTIAL);
            data = (const uint8_t *)p; mapped = true;
        }
        ::close(fd);
        return true;
    }
    ~MappedInput() { if (mapped) munmap((void *)data, size); }
    MappedInput() = default;
    MappedInput(const MappedInput &) = delete;
private:
    bool mapped = false;
    std::vector<uint8_t> own;
};

struct MappedOutput {                        // a file of up to `cap` bytes written through a shared mapping, cut to size at close
    uint8_t *data = nullptr;
    size_t cap = 0;
    bool open(const std::string &path, size_t capacity)
    {
        name = path; cap = capacity;
        fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return false;
        if (cap && ftruncate(fd, (off_t)cap) == 0) {
            void *p = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (p != MAP_FAILED) { data = (uint8_t *)p; mapped = true; return true; }
        }
        own.reset(new (std::nothrow) uint8_t[cap ? cap : 1]);       // fallback: plain buffer, written at close
        data = own.get();
        return data != nullptr;
    }
    bool close(size_t len)
    {
        bool ok = true;
        if (mapped) { ok = munmap(data, cap) == 0; mapped = false; if (ftruncate(fd, (off_t)len) != 0) ok = false; }
        else if (fd >= 0) {
            if (ftruncate(fd, 0) != 0) ok = false;
            size_t put = 0;
            while (ok && put < len) { ssize_t w = ::write(fd, data + put, len - put); if (w <= 0) ok = false; else put += (size_t)w; }
        }
        if (fd >= 0 && ::close(fd) != 0) ok = false;
        fd = -1; data = nullptr;
        return ok;
    }
    ~MappedOutput() { if (mapped) munmap(data, cap); if (fd >= 0) ::close(fd); }
    MappedOutput() = default;
    MappedOutput(const MappedOutput &) = delete;
private:
    std::string name;
    int fd = -1;
    bool mapped = false;
    std::unique_ptr<uint8_t[]> own;
};
