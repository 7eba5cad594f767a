// builder_main.cpp -- drop-in for the reference's step-1 tools, over libbfqhip.so:
//   gsufsort <in.fastq> --bwt --qs -o <OUT>                        (BFQzip.py:184)
//       -> <OUT>.bwt, <OUT>.bwt.qs   (terminator '#')
//   eGap <in.fastq> --em --mem <MB> --qs -o <OUT> --lcp --lbytes 1  (BFQzip_ext.py:172-177)
//       -> <OUT>.bwt (terminator byte 0), <OUT>.bwt.qs, <OUT>.<lbytes>.lcp
// Exit status 0 on success, 1 on any error (the drivers check it, BFQzip.py:328-336).
#include "cli_common.h"

int main(int argc, char **argv)
{
#ifdef BFQ_TOOL_EGAP
    const char *tool = "eGap";
    int term = 0;
#else
    const char *tool = "gsufsort";
    int term = '#';
#endif
    bfq_phase("start");
    std::string in, out;
    bool wantLcp = false;
    int lbytes = 1;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-o" || a == "--output") { if (++i < argc) out = argv[i]; }
        else if (a == "--mem" || a == "-m" || a == "--docs" || a == "-d") { ++i; }
        else if (a == "--lbytes") { if (++i < argc) lbytes = atoi(argv[i]); }
        else if (a == "--lcp") wantLcp = true;
        else if (a == "-v" || a == "--verbose") bfq_phase_enable(1);
        else if (a.size() && a[0] == '-') { /* --bwt --qs --em --rev ...: accepted, nothing to do */ }
        else if (in.empty()) in = a;
    }
    if (in.empty()) { fprintf(stderr, "%s: no input FASTQ\n", tool); return 1; }
    if (out.empty()) out = in;
    if (lbytes != 1 && lbytes != 2 && lbytes != 4) { fprintf(stderr, "%s: --lbytes must be 1, 2 or 4\n", tool); return 1; }
    // the file's bytes go to the GPU as they are (read into pinned staging by the library): records are indexed,
    // split and sorted there; the outputs are written straight from the staging buffers
    InFile buf;
    if (!buf.open(in)) { fprintf(stderr, "%s: cannot read %s\n", tool, in.c_str()); return 1; }
    bfq_params P;
    bfq_default_params(&P);
    P.piles = 1;                                 // one process, one collection: pile by pile in as little HBM as possible (what a process frees, the next one waits for)
    uint64_t n = 0, N = 0;
    OutFile bwt, qs, lcpf;
    bool ok = bwt.open(out + ".bwt") && qs.open(out + ".bwt.qs");
    if (ok && wantLcp) ok = lcpf.open(out + "." + std::to_string(lbytes) + ".lcp");
    if (!ok) { fprintf(stderr, "%s: cannot create outputs for %s\n", tool, out.c_str()); return 1; }
    // the outputs' pages are allocated and zeroed by helper threads from now on -- beside the lease, the start of the HIP
    // runtime, the upload and the sort (a failure here only means they are written the slow way)
    const uint64_t est = bfq_fastq_rows_estimate(buf.fd, buf.size), capRows = buf.size / 2 + 64;
    (void)bfq_output_prefault(bwt.fd, capRows, est);
    (void)bfq_output_prefault(qs.fd, capRows, est);
    if (wantLcp) (void)bfq_output_prefault(lcpf.fd, capRows * (uint64_t)lbytes, est * (uint64_t)lbytes);
    bfq_ctx *c = create_on_free_gpu(tool, &P);
    if (!c) return 1;
    int rc = bfq_fastq_build_ebwt_fd(c, buf.fd, buf.size, term, bwt.fd, qs.fd, wantLcp ? lcpf.fd : -1, lbytes, &n, &N);
    if (rc) { fprintf(stderr, "%s: %s: %s\n", tool, in.c_str(), bfq_last_error(c)); bfq_destroy(c); return 1; }
    bfq_phase("teardown");
    trace_kernel_times(c, tool);
    ok = bwt.close() && qs.close();
    ok = lcpf.close() && ok;
    bfq_destroy(c);
    if (!ok) { fprintf(stderr, "%s: cannot write outputs for %s\n", tool, out.c_str()); return 1; }
    bfq_phase_report(tool);
    printf("%s (bfqzip_amd/gfx950): %llu reads, %llu eBWT symbols -> %s.bwt, %s.bwt.qs%s\n", tool, (unsigned long long)N,
           (unsigned long long)n, out.c_str(), out.c_str(), wantLcp ? " (+lcp)" : "");
    return 0;
}
