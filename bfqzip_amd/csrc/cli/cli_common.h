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

// 4-line FASTQ records -> bases / quals back to back + offsets (lines 2 and 4)
static inline bool parse_fastq(const std::vector<uint8_t> &buf, std::vector<uint8_t> &bases, std::vector<uint8_t> &quals,
                               std::vector<uint64_t> &off, std::string &err)
{
    size_t n = buf.size(), p = 0;
    bases.clear(); quals.clear(); off.assign(1, 0);
    bases.reserve(n / 2); quals.reserve(n / 2);
    int line = 0;
    size_t seqLen = 0;
    while (p < n) {
        const uint8_t *nl = (const uint8_t *)memchr(buf.data() + p, '\n', n - p);
        size_t e = nl ? (size_t)(nl - buf.data()) : n;
        size_t le = e;
        if (le > p && buf[le - 1] == '\r') le--;
        if (line == 1) { bases.insert(bases.end(), buf.begin() + p, buf.begin() + le); seqLen = le - p; }
        else if (line == 3) {
            if (le - p != seqLen) { err = "len(DNA) != len(QS) in a record"; return false; }
            quals.insert(quals.end(), buf.begin() + p, buf.begin() + le);
            off.push_back(bases.size());
        }
        line = (line + 1) & 3;
        p = e + 1;
    }
    if (line != 0) { err = "number of lines is not a multiple of 4"; return false; }
    return true;
}
