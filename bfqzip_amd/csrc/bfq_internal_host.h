// bfq_internal_host.h -- host-only helpers of libbfqhip.so (bfq_host.cpp): no HIP types here.
#pragma once
#include <stdint.h>
#include <string>
#include "../../include/bfqzip_hip.h"

// The BFQ_* environment variables.  bfq_env(): read once per process (first use) -- tracing, lease, thread counts.
// A context keeps its own copy, refreshed by bfq_create() and bfq_set_params() only (test knobs change between calls).
struct BfqEnv {
    bool trace = false;             // BFQ_TRACE: phase timeline, transfer rates, workspace allocations on stderr
    int piles = 0;                  // BFQ_PILES=1/0/2: step 1 pile by pile always / never, 2: the capped mode always (overrides bfq_params.piles); unset: 0
    bool pilesSplit = false;        // BFQ_PILES_SPLIT: every pile once more by its second symbol (test knob)
    bool noOverlap = false;         // BFQ_NO_OVERLAP: inversion and device -> host copies one after the other
    bool noLengthGuess = false;     // BFQ_NO_LENGTH_GUESS: always count read lengths by LF walks
    bool posMode = false;           // BFQ_POSMODE=1: steps 3-4 without LF table (position mode)
    int invertThreads = 0;          // BFQ_INVERT_THREADS: workgroup size of k_invert (0: default)
    int ioThreads = 0;              // BFQ_IO_THREADS: staging workers (0: by core count)
    int prefaultThreads = -1;       // BFQ_PREFAULT_THREADS: helpers that fault in output mappings (-1: by core count)
    unsigned long long hugeCap = 0; // BFQ_HUGE_CAP: slot budget of the huge-segment rounds (test knob)
    unsigned long long wsCap = 0;   // BFQ_WS_CAP: upper bound of the device workspace in bytes (suffixes K/M/G), 0: none
    int device = -1;                // BFQ_DEVICE: the GPU the one-shot tools use (-1: first free one, by lease)
    int fakeDevices = 0;            // BFQ_FAKE_DEVICES=n: pretend n GPUs (slot k -> device k mod the real count): lease tests on one GPU
    bool lease = true;              // BFQ_LEASE=0: no lease files (the caller places the tools itself)
    std::string leaseDir;           // BFQ_LEASE_DIR: where the lock files live (default /dev/shm, else /tmp)
    bool rsPerm = false;            // BFQ_RS_PERM=1: the radix passes take their blocks spread by a coprime stride instead of in order (placement experiments: no effect)
    int wsVmmMib = 0;               // BFQ_WS_VMM=<MiB>: the workspace as physically contiguous chunks of that size mapped into one range (hipMemCreate / hipMemMap)
    bool wsContig = false;          // BFQ_WS_CONTIG=1: ask for a physically contiguous workspace (hipDeviceMallocContiguous), plain hipMalloc if refused
    unsigned long long abPad = 0;   // BFQ_AB_PAD: bytes left free between the two sort-record buffers (placement experiments)
    char abOrder[8] = {0};          // BFQ_AB_ORDER=<permutation of 0123>: the order of A.w12, A.w0, B.w0, B.w12 in the arena (placement experiments)
    bool abSwap = false;            // BFQ_AB_SWAP=1: the second record buffer below the first (placement experiments)
    bool compact = false;           // BFQ_COMPACT=1: steps 2-4 on a given eBWT + LCP without the LF table whatever the cap (k_compact.hip; test knob)
    bool keyFusion = false;         // BFQ_KEY_FUSION=1: the sort's records are made by its first scatter pass from the text (slower: bfq_api.hip)
    int dnaStatic = 0;              // BFQ_DNA_STATIC=1: read-order DNA through the static BFQRANS2 container as well (4x faster, 3x larger)
    int dnacK = 0, dnacH = 0, dnacW = 0, dnacSkip = -1;   // BFQ_DNAC_K / _H / _W / _TSKIP: the BFQDNAC1 container's parameters (the header carries them; defaults in k_dnac.hip)
    unsigned long long compactRing = 0; // BFQ_COMPACT_RING: entries of the interval refinement's ring queue there (default: what the cap leaves; small values test the chunked levels / the move to host memory)
    unsigned long long compactWin = 0;  // BFQ_COMPACT_WIN: rows of the LCP file in flight there (default 64 Mi; small values test the windowing)
    bool prefaultPause = false;     // BFQ_PREFAULT_PAUSE=1: output files are not allocated while an input file is being read (default: both at once)
    bool noOutmap = false;          // BFQ_NO_OUTMAP: the tools write their outputs with pwrite instead of through a mapping (test knob)
    int invertNt = 1;               // BFQ_INVERT_NT=0: plain instead of nontemporal LF-table loads in k_invert
};
BfqEnv bfq_env_read();
const BfqEnv &bfq_env();
int bfq_cpu_budget();            // CPUs this process may keep busy (cgroup quota, affinity, hardware)

// output file mapped for writing, pre-faulted in the background
struct bfq_outmap;
bfq_outmap *bfq_outmap_open(int fd, uint64_t map_len, uint64_t prefault_len);
bfq_outmap *bfq_outmap_take(int fd, uint64_t min_len);     // a mapping registered by bfq_output_prefault(), or nullptr
char *bfq_outmap_ptr(bfq_outmap *m);
uint64_t bfq_outmap_len(bfq_outmap *m);
void bfq_outmap_extend(bfq_outmap *m, uint64_t prefault_len);
bool bfq_outmap_close(bfq_outmap *m, uint64_t final_len);
bool bfq_outmap_ensure(bfq_outmap *m, uint64_t off, uint64_t len);   // before copying into [off, off + len); false: use pwrite on bfq_outmap_fd()
int bfq_outmap_fd(bfq_outmap *m);
