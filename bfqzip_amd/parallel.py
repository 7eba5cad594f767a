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
