"""Block sharding of a read collection across GPUs: the reference's own split rule.

BFQzip_parallel.py:288-323 (split_fastq): size_block = num_reads // t,
num_blocks = num_reads // size_block; blocks 0..num_blocks-2 take size_block
consecutive reads, the LAST block takes all remaining reads (so num_blocks may
exceed t).  Paired mode (split_fastq_2, :325-360): block k of file 2 (same rule on
file 2) is appended to block k of file 1; after inversion the first
len(block k of file 1) reads go to OUT_1, the rest to OUT_2 (:153-172).
Every block is a fully independent run of the hot path (own eBWT, clusters,
inversion); the merge is an ordered concatenation (`cat`, :174-177).

Multi-GPU mapping: one process per GPU (torch.distributed; backend nccl = RCCL
on GPUs, gloo in the CPU tests), blocks dealt round-robin to ranks, no data-path
collective; the only exchange is the ordered gather of the per-block outputs
(variable-length byte buffers) to rank 0.
"""
import numpy as np


def split_blocks(num_reads, t):
    """[(first_read, end_read)] per block, exactly as split_fastq() cuts them."""
    if num_reads <= 0:
        return []
    t = max(1, int(t))
    size_block = num_reads // t
    if size_block == 0:                       # more threads than reads: the reference divides by zero; one block
        return [(0, num_reads)]
    num_blocks = num_reads // size_block
    out = []
    for b in range(num_blocks):
        s = b * size_block
        e = num_reads if b == num_blocks - 1 else s + size_block
        out.append((s, e))
    return out


def blocks_of_rank(num_blocks, rank, world):
    return list(range(rank, num_blocks, world))


def slice_reads(bases, quals, roff, s, e):
    """Reads [s,e) of a collection as its own collection."""
    lo, hi = int(roff[s]), int(roff[e])
    return bases[lo:hi], quals[lo:hi], (roff[s:e + 1] - roff[s]).astype(np.uint64)


def paired_blocks(c1, c2, t):
    """Blocks for -p: block k = reads of file-1 block k followed by reads of file-2 block k.
    Returns [(bases, quals, roff, n_reads_from_file1)]."""
    b1, b2 = split_blocks(len(c1[2]) - 1, t), split_blocks(len(c2[2]) - 1, t)
    out = []
    for k in range(max(len(b1), len(b2))):
        parts = []
        n1 = 0
        if k < len(b1):
            parts.append(slice_reads(*c1, *b1[k])); n1 = b1[k][1] - b1[k][0]
        if k < len(b2):
            parts.append(slice_reads(*c2, *b2[k]))
        bases = np.concatenate([p[0] for p in parts]); quals = np.concatenate([p[1] for p in parts])
        roffs = [parts[0][2]]
        for p in parts[1:]:
            roffs.append(p[2][1:] + roffs[-1][-1])
        out.append((bases, quals, np.concatenate(roffs).astype(np.uint64), n1))
    return out


def run_blocks(run_block, bases, quals, roff, t, dist=None, device=None):
    """Process a collection block-wise.  run_block(bases, quals, roff) -> (out_bases, out_quals).
    With torch.distributed initialised (dist), blocks are dealt round-robin to the ranks and
    rank 0 returns the ordered concatenation (other ranks return None)."""
    blocks = split_blocks(len(roff) - 1, t)
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    mine = {}
    for k in blocks_of_rank(len(blocks), rank, world):
        bb, bq, br = slice_reads(bases, quals, roff, *blocks[k])
        ob, oq = run_block(bb, bq, br)
        mine[k] = (np.ascontiguousarray(ob, np.uint8), np.ascontiguousarray(oq, np.uint8))
    if dist is None or world == 1:
        ks = sorted(mine)
        return (np.concatenate([mine[k][0] for k in ks]) if ks else np.zeros(0, np.uint8),
                np.concatenate([mine[k][1] for k in ks]) if ks else np.zeros(0, np.uint8))
    return gather_blocks(mine, blocks, roff, dist, device)


def gather_blocks(mine, blocks, roff, dist, device=None):
    """Ordered gather of per-block byte buffers to rank 0: sizes are known from the split
    (inversion keeps read lengths), payloads travel as one padded uint8 tensor per rank."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    sizes = [int(roff[e]) - int(roff[s]) for s, e in blocks]
    per_rank = [sum(sizes[k] for k in blocks_of_rank(len(blocks), r, world)) for r in range(world)]
    cap = max(per_rank) if per_rank else 0
    buf = torch.zeros(2 * cap if cap else 1, dtype=torch.uint8, device=dev)
    o = 0
    for k in sorted(mine):
        n = sizes[k]
        buf[o:o + n] = torch.from_numpy(mine[k][0]).to(dev)
        buf[cap + o:cap + o + n] = torch.from_numpy(mine[k][1]).to(dev)
        o += n
    got = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, got, dst=0)
    if rank != 0:
        return None
    total = int(roff[-1])
    ob = np.empty(total, np.uint8); oq = np.empty(total, np.uint8)
    for r in range(world):
        g = got[r].cpu().numpy()
        o = 0
        for k in blocks_of_rank(len(blocks), r, world):
            n = sizes[k]
            lo = int(roff[blocks[k][0]])
            ob[lo:lo + n] = g[o:o + n]
            oq[lo:lo + n] = g[cap + o:cap + o + n]
            o += n
    return ob, oq


def main(argv=None):
    """`torchrun --nproc-per-node G -m bfqzip_amd.parallel in.fastq [in2.fastq -p] -o OUT -t n [--headers] [...]`

    The multi-GPU counterpart of `BFQzip_parallel.py in.fastq -o OUT -t n -0`: same block split, one block per
    GPU at a time, outputs merged in block order into OUT.fq (or OUT_1.fq / OUT_2.fq with -p).  Without torchrun it
    runs all blocks on GPU 0."""
    import argparse, os
    import torch
    from . import api, fastq
    ap = argparse.ArgumentParser(prog="bfqzip_amd.parallel")
    ap.add_argument("input"); ap.add_argument("input2", nargs="?")
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("-t", "--threads", type=int, default=1, help="number of blocks (BFQzip_parallel.py -t)")
    ap.add_argument("-p", "--paired", action="store_true")
    ap.add_argument("--headers", action="store_true")
    ap.add_argument("--m2", action="store_true", help="also write OUT.fq.dna and OUT.fq.qs (BFQzip.py:231-249)")
    ap.add_argument("--m3", action="store_true", help="--m2 plus OUT.h (BFQzip.py:195-201)")
    ap.add_argument("-T", dest="k", type=int, default=16); ap.add_argument("-Q", dest="v", default=">")
    ap.add_argument("--M", type=int, default=2); ap.add_argument("--B", type=int, default=0)
    a = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("BFQ_BACKEND", "nccl")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    rank = dist.get_rank() if dist else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local) if (dist and dist.get_backend() == "nccl") else None
    eng = api.Engine(local, k=a.k, m=5, v=ord(a.v), M=a.M, B=a.B)          # -m 5: what BFQzip.py passes (BFQzip.py:215)

    def run(b, q, r):
        ob, oq, st = eng.run_reads(b, q, r)
        return ob, oq

    c1 = fastq.read_fastq(a.input)
    if a.paired and a.input2:
        c2 = fastq.read_fastq(a.input2)
        blocks = paired_blocks(c1[:3], c2[:3], a.threads)
        o1, o2 = [], []
        for k, (bb, bq, br, n1) in enumerate(blocks):                       # paired: blocks stay on this rank's GPU
            if k % world != rank:
                continue
            ob, oq = run(bb, bq, br)
            cut = int(br[n1])
            o1.append((k, fastq.format_fastq(ob[:cut], oq[:cut], br[:n1 + 1])))
            o2.append((k, fastq.format_fastq(ob[cut:], oq[cut:], (br[n1:] - br[n1]).astype(np.uint64))))
        if dist:
            g1 = [None] * world; g2 = [None] * world
            dist.all_gather_object(g1, o1); dist.all_gather_object(g2, o2)
            o1 = [x for g in g1 for x in g]; o2 = [x for g in g2 for x in g]
        if rank == 0:
            open(a.out + "_1.fq", "wb").write(b"".join(t for _, t in sorted(o1)))
            open(a.out + "_2.fq", "wb").write(b"".join(t for _, t in sorted(o2)))
    else:
        b, q, r, h = c1
        res = run_blocks(run, b, q, r, a.threads, dist=dist, device=dev)
        if rank == 0:
            open(a.out + ".fq", "wb").write(fastq.format_fastq(res[0], res[1], r, h if a.headers else None))
            if a.m2 or a.m3:
                open(a.out + ".fq.dna", "wb").write(fastq.format_lines(res[0], r))
                open(a.out + ".fq.qs", "wb").write(fastq.format_lines(res[1], r))
            if a.m3:
                open(a.out + ".h", "wb").write(b"".join(x + b"\n" for x in h))
    eng.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
