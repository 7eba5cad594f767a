// bsc_main.cpp -- drop-in for the command line BFQzip.py hands every output stream to in step 5b
// (`external/libbsc/bsc e <stream> <stream>.bsc -T`, BFQzip.py:23,265-275):
//     bsc e INPUT OUTPUT [options]     compress   (libbsc's options are accepted and ignored: the codec has none)
//     bsc d INPUT OUTPUT [options]     decompress
//     bsc x ROWS.z QS.z OUT.dna OUT.qs     (not a libbsc command) eBWT-domain containers (bfq_fastq_job.compress_streams = 2 / 3)
//                                          back to the line streams OUT.fq.dna / OUT.fq.qs
// The container is this project's BFQRANS2 (include/bfqzip_hip.h, oracle/bfq_codec_ref.c), written by the GPU codec;
// it is NOT libbsc's format (libbsc is an empty submodule of the reference tree).  Exit status 0 on success.
#include "cli_common.h"

int main(int argc, char **argv)
{
    if (argc >= 6 && !strcmp(argv[1], "x")) {
        std::vector<uint8_t> bz, qz;
        if (!read_file(argv[2], bz) || !read_file(argv[3], qz)) { fprintf(stderr, "bsc: cannot read %s / %s\n", argv[2], argv[3]); return 1; }
        if (bz.size() < 40 || memcmp(bz.data(), "BFQEBWT1", 8)) { fprintf(stderr, "bsc: %s is not a BFQEBWT1 stream\n", argv[2]); return 1; }
        uint64_t rows = 0, symLen = 0;
        memcpy(&rows, bz.data() + 8, 8);
        memcpy(&symLen, bz.data() + 32, 8);
        // the header is not trusted: the row count must be what the containers inside say they decode to, before anything
        // is sized by it
        if (symLen > bz.size() - 40 || bfq_stream_raw_len(bz.data() + 40, symLen) != (int64_t)rows ||
            bfq_stream_raw_len(bz.data() + 40 + symLen, bz.size() - 40 - symLen) != (int64_t)rows ||
            bfq_stream_raw_len(qz.data(), qz.size()) != (int64_t)rows) {
            fprintf(stderr, "bsc: %s / %s: damaged BFQEBWT1 stream\n", argv[2], argv[3]);
            return 1;
        }
        bfq_params P;
        bfq_default_params(&P);
        bfq_ctx *c = create_on_free_gpu("bsc", &P);
        if (!c) return 1;
        std::vector<uint8_t> dna(rows + 1), qs(rows + 1);
        uint64_t sl = 0, nr = 0;
        const int rc = bfq_stream_ebwt_decode(c, bz.data(), bz.size(), qz.data(), qz.size(), dna.data(), qs.data(), rows, &sl, &nr);
        if (rc) { fprintf(stderr, "bsc: %s\n", bfq_last_error(c)); bfq_destroy(c); return 1; }
        bfq_destroy(c);
        if (!write_file(argv[4], dna.data(), sl) || !write_file(argv[5], qs.data(), sl)) { fprintf(stderr, "bsc: cannot write the outputs\n"); return 1; }
        printf("%llu reads, %llu bytes per stream\n", (unsigned long long)nr, (unsigned long long)sl);
        return 0;
    }
    if (argc < 4 || (strcmp(argv[1], "e") && strcmp(argv[1], "d"))) {
        fprintf(stderr, "usage: %s e|d INPUT OUTPUT [options]   |   %s x ROWS.z QS.z OUT.dna OUT.qs\n", argv[0], argv[0]);
        return 1;
    }
    const bool enc = !strcmp(argv[1], "e");
    std::vector<uint8_t> in;
    if (!read_file(argv[2], in)) { fprintf(stderr, "bsc: cannot read %s\n", argv[2]); return 1; }
    bfq_params P;
    bfq_default_params(&P);
    bfq_ctx *c = create_on_free_gpu("bsc", &P);
    if (!c) return 1;
    uint64_t cap;
    if (enc) cap = bfq_stream_bound(in.size());
    else {
        const int64_t raw = bfq_stream_raw_len(in.data(), in.size());
        if (raw < 0) { fprintf(stderr, "bsc: %s is not a BFQRANS2 / BFQDNAC1 / BFQLINE1 stream\n", argv[2]); bfq_destroy(c); return 1; }
        cap = (uint64_t)raw;
    }
    std::vector<uint8_t> out(cap ? cap : 1);
    uint64_t got = 0;
    const int rc = enc ? bfq_stream_compress(c, in.data(), in.size(), out.data(), out.size(), &got)
                       : bfq_stream_decompress(c, in.data(), in.size(), out.data(), out.size(), &got);
    if (rc) { fprintf(stderr, "bsc: %s\n", bfq_last_error(c)); bfq_destroy(c); return 1; }
    bfq_destroy(c);
    if (!write_file(argv[3], out.data(), got)) { fprintf(stderr, "bsc: cannot write %s\n", argv[3]); return 1; }
    if (enc) printf("%s compressed %llu into %llu (GPU context model + rANS: BFQDNAC1 / BFQRANS2 / BFQLINE1)\n", argv[2], (unsigned long long)in.size(), (unsigned long long)got);
    else printf("%s decompressed %llu into %llu\n", argv[2], (unsigned long long)in.size(), (unsigned long long)got);
    return 0;
}
