// bfq_main.cpp -- drop-in for src_int_mem/bfq_int and src_ext_mem/bfq_ext over libbfqhip.so.
//   bfq_int -e OUT.bwt -q OUT.bwt.qs -o OUT.fq -m 5 [-k T] [-v ord(Q)] [-H OUT.h]      (BFQzip.py:215-222)
//   bfq_ext -e .. -q .. -a OUT.1.lcp -o OUT -l 250 -s 0 -m 5 [-k][-v][-H]             (BFQzip_ext.py:208-214)
// Flags, defaults and sentinel substitution follow bfq_int.cpp:883-935 (getopt
// "he:q:o:k:m:v:f:t:s:DVH:") and bfq_ext.cpp:969; usage / missing-file errors print
// the help and exit 0 like the reference (bfq_int.cpp:132,937-955); a forbidden
// eBWT symbol exits 1 (dna_string_n.hpp:87-93).  M and B are compile-time knobs
// in the reference (-DM/-DB, src_int_mem/Makefile:13-23): here `make M=.. B=..`
// sets the defaults and the environment variables BFQ_M / BFQ_B override them.
#include <unistd.h>
#include "cli_common.h"

#ifndef BFQ_DEFAULT_M
#define BFQ_DEFAULT_M 2
#endif
#ifndef BFQ_DEFAULT_B
#define BFQ_DEFAULT_B 0
#endif
#ifdef BFQ_TOOL_EXT
static const char *TOOL = "bfq_ext";
static const char *OPTS = "he:q:o:a:l:k:m:v:f:t:s:T:H:DV";
#else
static const char *TOOL = "bfq_int";
static const char *OPTS = "he:q:o:k:m:v:f:t:s:DVH:";
#endif

static void help()
{
    printf("%s [options]\nOptions:\n"
           "-h          Print this help.\n"
           "-e <arg>    Input eBWT file (A,C,G,T,#) of DNA (REQUIRED).\n"
           "-q <arg>    Qualities permuted according to the DNA's ebwt (REQUIRED).\n"
#ifdef BFQ_TOOL_EXT
           "-a <arg>    LCP array file (REQUIRED).\n"
           "-l <arg>    Maximum read length.\n"
           "-T <arg>    Threads (ignored: the GPU does the work).\n"
#endif
           "-o <arg>    Output fastq (REQUIRED).\n"
           "-k <arg>    Minimum LCP required in clusters. Default: 16.\n"
           "-m <arg>    Minimum length of cluster to be processed. Default: 2.\n"
           "-v <arg>    Quality score for constant replacement (if M=2). Default: 29.\n"   /* the phred value of the default '>': bfq_int.cpp:122 prints (int)default_value_def - 33 */
           "-f <arg>    Percentage threshold for frequent bases in clusters. Default: 40.\n"
           "-t <arg>    Quality score threshold for trusted bases. Default: 20.\n"
           "-s <arg>    ASCII value of terminator character. Default: 35 (#).\n"
           "-H <arg>    List of original headers.\n", TOOL);
    exit(0);
}

int main(int argc, char **argv)
{
    bfq_phase("start");
    std::string in_dna, in_qual, in_lcp, output, titles;
    int K = -1, m = 0, v = 0, f = 0, t = -1, term = '#';
    bool headers = false, verbose = false;
    if (argc < 3) help();
    int opt;
    while ((opt = getopt(argc, argv, OPTS)) != -1) {
        switch (opt) {
        case 'h': help(); break;
        case 'e': in_dna = optarg; break;
        case 'q': in_qual = optarg; break;
        case 'o': output = optarg; break;
        case 'a': in_lcp = optarg; break;
        case 'l': case 'T': break;
        case 'k': K = atoi(optarg); break;
        case 'm': m = atoi(optarg); break;
        case 'v': v = atoi(optarg); break;
        case 'f': f = atoi(optarg); break;
        case 't': t = atoi(optarg); break;
        case 's': term = atoi(optarg); break;
        case 'D': break;                       // debug dump: unusable in the reference
        case 'V': verbose = true; bfq_phase_enable(1); break;
        case 'H': titles = optarg; headers = true; break;
        default: help(); return -1;
        }
    }
    bfq_params P;
    bfq_default_params(&P);
    P.K = K == -1 ? 16 : K;
    P.m = m == 0 ? 2 : m;
    P.v = (char)v == '\0' ? '>' : (signed char)v;
    P.t = t == -1 ? 20 : t;
    P.f = f == 0 ? 40 : f;
    P.term = term & 0xFF;
    P.M = BFQ_DEFAULT_M; P.B = BFQ_DEFAULT_B;
    if (getenv("BFQ_M")) P.M = atoi(getenv("BFQ_M"));
    if (getenv("BFQ_B")) P.B = atoi(getenv("BFQ_B"));
#ifdef BFQ_TOOL_EXT
    P.ext = 1;
    bool needLcp = true;
#else
    bool needLcp = false;
#endif
    if (in_dna.empty() || in_qual.empty() || output.empty() || (needLcp && in_lcp.empty())) help();
    if (!file_exists(in_dna)) { printf("Error: could not find file %s.\n\n", in_dna.c_str()); help(); }
    if (!file_exists(in_qual)) { printf("Error: could not find file %s\n\n", in_qual.c_str()); help(); }
    if (needLcp && !file_exists(in_lcp)) { printf("Error: could not find file %s\n\n", in_lcp.c_str()); help(); }
    if (headers && !file_exists(titles)) { printf("Error: could not find file %s.\n\n", titles.c_str()); help(); }
#ifdef BFQ_TOOL_EXT
    output += ".fq";                            // decode.cpp:279 appends the extension itself
#endif
    printf("Running %s (bfqzip_amd/gfx950)...\n\tMode: %d", TOOL, P.M);
    if (P.M == 2) printf("\treplacing QS with symbol: %c", (char)P.v);
    printf("\n\tIllumina 8-level binning: %d\n\tK: %d\n\tm: %d\n\tFrequency threshold: %d%%\n\nOutput fastq file: %s\n\n",
           P.B, P.K, P.m, P.f, output.c_str());

    InFile bwt, qs, lcp, hdr;
    if (!bwt.open(in_dna) || !qs.open(in_qual)) { fprintf(stderr, "%s: cannot read inputs\n", TOOL); return 1; }
    if (qs.size != bwt.size) { fprintf(stderr, "%s: eBWT and QS lengths differ (bfq_int.cpp:649)\n", TOOL); return 1; }
    if (needLcp) {
        if (!lcp.open(in_lcp)) { fprintf(stderr, "%s: cannot read %s\n", TOOL, in_lcp.c_str()); return 1; }
        if (lcp.size % (bwt.size ? bwt.size : 1) != 0 || (bwt.size && lcp.size / bwt.size != 1 &&
            lcp.size / bwt.size != 2 && lcp.size / bwt.size != 4)) {
            fprintf(stderr, "%s: LCP file size does not match the eBWT\n", TOOL); return 1;
        }
    }
    uint64_t n = bwt.size;
    // the FASTQ text (header line verbatim from -H, else "@"; bases; "+"; qualities -- bfq_int.cpp:797-810)
    // is laid out on the GPU and written straight from the library's staging buffers
    if (headers && !hdr.open(titles)) { fprintf(stderr, "%s: cannot read %s\n", TOOL, titles.c_str()); return 1; }
    OutFile outText;
    if (!outText.open(output)) { perror("invert"); return 1; }
    // at least 2 n bytes of it are certain: their pages are faulted in by helper threads from now on (include/bfqzip_hip.h)
    (void)bfq_output_prefault(outText.fd, 6 * n + hdr.size + 4096, 2 * n);
    bfq_ctx *c = create_on_free_gpu(TOOL, &P);
    if (!c) return 1;
    uint64_t outLen = 0;
    bfq_stats st;
    int lb = (needLcp && n) ? (int)(lcp.size / n) : 0;
    int rc = bfq_smooth_invert_fastq_fd(c, bwt.fd, qs.fd, needLcp ? lcp.fd : -1, lb, n, headers ? hdr.fd : -1, hdr.size, outText.fd, &outLen, &st);
    if (rc) {
        fprintf(stderr, "%s: %s\n", TOOL, bfq_last_error(c));
        bfq_destroy(c);
        return 1;
    }
    bfq_phase("teardown");
    trace_kernel_times(c, TOOL);
    const bool closed = outText.close();
    bfq_destroy(c);
    if (!closed) { perror("invert"); return 1; }
    bfq_phase_report(TOOL);
    const uint64_t N = st.n_reads;
    printf("Number of reads: %llu\n", (unsigned long long)N);

    // bfq_int.cpp:1004-1019
    double nb = (double)(n - N), nc = (double)st.num_clust;
    printf("**** Cluster statistics ****\nTot: %llu\n", (unsigned long long)st.num_clust);
    printf("%llu (%g%%) bases fall inside clusters\n", (unsigned long long)st.bases_inside, nb ? 100.0 * st.bases_inside / nb : 0.0);
    printf("Discarded: %llu(%g%%)\n", (unsigned long long)st.num_clust_discarded, nc ? 100.0 * st.num_clust_discarded / nc : 0.0);
    printf("Ambiguous discarded: %llu(%g%%)\n", (unsigned long long)st.num_clust_amb_discarded, nc ? 100.0 * st.num_clust_amb_discarded / nc : 0.0);
    printf("Processed: %llu(%g%%)\n", (unsigned long long)st.num_clust_mod, nc ? 100.0 * st.num_clust_mod / nc : 0.0);
    printf("Clusters with only one symbol: %llu(%g%%)\n\n", (unsigned long long)st.num_clust_alleq, nc ? 100.0 * st.num_clust_alleq / nc : 0.0);
    printf("**** Quality statistics ****\n%llu/%llu qualities have been modified\n***********************\n\n",
           (unsigned long long)st.qs_smoothed, (unsigned long long)(n - N));
    printf("**** Bases statistics ****\n%llu/%llu bases have been modified\n***********************\n\n",
           (unsigned long long)st.modified, (unsigned long long)(n - N));
    (void)verbose;
    return 0;
}
