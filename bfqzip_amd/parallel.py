"""Block sharding of a FASTQ file across GPUs: the reference's own split / merge rule, on bytes.

BFQzip_parallel.py:288-323 (split_fastq): size_block = num_reads // t, num_blocks = num_reads // size_block;
blocks 0..num_blocks-2 take size_block consecutive reads, the LAST block takes all remaining reads (so
num_blocks may exceed t).  Paired mode (split_fastq_2, :325-360): block k of file 2 (same rule on file 2) is
appended to block k of file 1; after the run the first lines(block k of file 1) output lines go to OUT_1, the
rest to OUT_2 (:153-172).  Every block is a fully independent run of the hot path (`BFQzip.py <block> --rebuild
-0 [--headers]`, :277-285: own eBWT, clusters, inversion); the merge is an ordered concatenation (`cat`, :174-177).

Here: one process per GPU (torch.distributed; backend nccl = RCCL on GPUs, gloo in the CPU tests).  Nothing is
done per read on the host:
  * the input files are memory-mapped; every rank counts the newlines of ITS share of the bytes (host threads
    in libbfqhip.so), the per-chunk counts are all-gathered (a few thousand integers), and block boundaries
    are located by line number inside single 1 MiB chunks;
  * a block = 1 byte range (2 with -p: the mate block appended) handed to Engine.fastq_job, which parses,
    runs the whole path and formats the FASTQ text / the --m2/--m3 streams on the GPU;
  * blocks are dealt round-robin; after every round of `world` blocks the output sizes (6 integers per block)
    are all-gathered and each rank writes its block's bytes at their final offsets (pwrite) -- there is no
    data-path collective and no gather of payload bytes.
"""
import os
import sys
import numpy as np

CHUNK = 1 << 20


def split_blocks(num_reads, t):
    """[(first_read, end_read)] per block, exactly as split_fastq() cuts them (t == 0: one block, :302-303)."""
    if num_reads <= 0:
        return []
    t = int(t)
    size_block = num_reads if t <= 0 else num_reads // t
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


class Comm:
    """The few integers the ranks exchange; single process when torch.distributed is not initialised."""

    def __init__(self, dist=None, device=None):
        self.dist, self.device = dist, device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1

    def all_gather_i64(self, vec):
        """vec: int64 array of the same length on every rank -> array [world, len]."""
        vec = np.ascontiguousarray(vec, np.int64)
        if self.dist is None or self.world == 1:
            return vec[None, :].copy()
        import torch
        dev = self.device if self.device is not None else torch.device("cpu")
        mine = torch.from_numpy(vec).to(dev)
        got = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(got, mine)
        return np.stack([g.cpu().numpy() for g in got])

    def barrier(self):
        if self.dist is not None and self.world > 1:
            self.dist.barrier()

    def _direct(self, t):
        return self.dist is not None and self.dist.get_backend() == "nccl" and t.is_cuda

    def broadcast_(self, t, src):
        """In-place broadcast of a tensor (RCCL on GPU tensors; through the host under gloo)."""
        if self.dist is None or self.world == 1 or t.numel() == 0:
            return
        if self._direct(t) or not t.is_cuda:
            self.dist.broadcast(t, src)
        else:
            h = t.cpu()
            self.dist.broadcast(h, src)
            t.copy_(h)

    def all_reduce_sum_(self, t):
        if self.dist is None or self.world == 1 or t.numel() == 0:
            return
        if self._direct(t) or not t.is_cuda:
            self.dist.all_reduce(t)
        else:
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)


def pwrite_all(fd, data, offset, host=None):
    """`data` into the open file at `offset`: through the library's threaded writer (bfq_file_put: fallocate + shared
    mapping, several threads -- a 9.5 GB output takes a Python os.pwrite 1.6 s) when the engine's host helpers offer it,
    else os.pwrite until everything is written (one call moves at most 0x7ffff000 bytes on Linux)."""
    put = getattr(host, "file_put", None) if host is not None else None
    if put is not None and len(data) >= (1 << 20):
        put(fd, data, offset)
        return
    mv = memoryview(data).cast("B")
    done = 0
    while done < len(mv):
        done += os.pwrite(fd, mv[done:done + (1 << 30)], offset + done)


def map_file(path):
    """Read-only uint8 view of a file (nothing is read until touched)."""
    if os.path.getsize(path) == 0:
        return np.zeros(0, np.uint8)
    return np.memmap(path, dtype=np.uint8, mode="r")


class TextIndex:
    """Line index of a memory-mapped text, built by all ranks together: rank r counts the newlines of the r-th
    share of the chunks; cum[i] = newlines before chunk i."""

    def __init__(self, buf, comm, count_fn, nth_fn, chunk=CHUNK):
        self.buf, self.chunk, self.nth_fn = buf, chunk, nth_fn
        n = len(buf)
        nch = (n + chunk - 1) // chunk
        per = (nch + comm.world - 1) // comm.world if nch else 0
        lo, hi = min(nch, comm.rank * per), min(nch, (comm.rank + 1) * per)
        mine = np.zeros(per, np.int64)
        if hi > lo:
            mine[:hi - lo] = count_fn(buf[lo * chunk:min(n, hi * chunk)], chunk).astype(np.int64)
        allc = comm.all_gather_i64(mine).reshape(-1)[:nch] if per else np.zeros(0, np.int64)
        self.cum = np.zeros(nch + 1, np.int64)
        np.cumsum(allc, out=self.cum[1:])
        self.num_lines = int(self.cum[-1]) + (1 if n and buf[n - 1] != 10 else 0)

    def line_start(self, k):
        """Byte offset at which line k (0-based) starts; k == num_lines -> end of the text."""
        n = len(self.buf)
        if k <= 0:
            return 0
        if k >= self.num_lines:
            return n
        j = k - 1                                                   # the newline that ends line k-1
        ci = int(np.searchsorted(self.cum, j, side="right")) - 1
        b = ci * self.chunk
        off = self.nth_fn(self.buf[b:min(n, b + self.chunk)], j - int(self.cum[ci]))
        if off < 0:
            raise RuntimeError("line index inconsistent with the text (file changed while mapped?)")
        return b + off + 1


def byte_blocks(index, t):
    """[(byte0, byte1, n_reads)] per block of split_fastq(); num_reads = num_lines // 4 (:298)."""
    if index.num_lines % 4:
        raise ValueError("FASTQ: number of lines is not a multiple of 4")
    out = []
    for s, e in split_blocks(index.num_lines // 4, t):
        out.append((index.line_start(4 * s), index.line_start(4 * e), e - s))
    return out


def output_names(inputs, out, paired):
    """Merged FASTQ names as BFQzip_parallel.py:142-153 builds them (+ the stream names of BFQzip.py:192-251)."""
    root1, ext1 = os.path.splitext(inputs[0])
    if not out:
        fq = [root1 + ".cat" + ext1]
        base = [root1 + ".cat"]
        if paired:
            root2, ext2 = os.path.splitext(inputs[1])
            fq.append(root2 + ".cat" + ext2); base.append(root2 + ".cat")
    elif paired:
        fq = [out + "_1" + ext1, out + "_2" + ext1]
        base = [out + "_1", out + "_2"]
    else:
        fq = [out + ext1]
        base = [out]
    return [{"fastq": f, "dna": f + ".dna", "qs": f + ".qs", "hdr": b + ".h"} for f, b in zip(fq, base)]


KINDS = ("fastq", "dna", "qs", "hdr")


def _pinned_outputs(kinds, cap):
    """Page-locked output buffers of `cap` bytes for the wanted kinds (direct DMA instead of the staging pipeline)."""
    from . import api
    pins = {k: api.PinnedBuffer(cap) for k in kinds}
    return pins, {k: p.array for k, p in pins.items()}


def run_files(eng, comm, inputs, t, names, paired=False, headers=False, want_fastq=True, want_streams=False,
              want_hdr=False, out_bufs=None, log=None, compress=False, pinned=False):
    """The whole multi-GPU job.  eng: Engine-like (fastq_job, text_line_counts/text_nth_newline via `eng.host`).
    Returns per-rank totals {"blocks", "reads", "bases", "stats"} (stats summed over this rank's blocks).
    compress: step 5 too (BFQzip.py:253-275) -- every block's share of every output goes through the stream codec
    (eng.stream_compress) and the files `<name>.bsc` hold one BFQRANS2 container per block, in block order
    (`bsc d` / stream_decompress read them back as one stream)."""
    host = eng.host
    bufs = [map_file(p) for p in inputs]
    idx = [TextIndex(b, comm, host.text_line_counts, host.text_nth_newline) for b in bufs]
    blocks = [byte_blocks(i, t) for i in idx]
    nblocks = len(blocks[0])                                         # the blocks of file 1 drive the run (:97-119)
    kinds = [k for k, w in zip(KINDS, (want_fastq, want_streams, want_streams, want_hdr)) if w]
    nout = 2 if paired else 1
    if compress:
        names = [{k: v + ".bsc" for k, v in nm.items()} for nm in names]
    if comm.rank == 0:                                               # create / truncate the outputs once
        for o in range(nout):
            for k in kinds:
                open(names[o][k], "wb").close()
    comm.barrier()
    fds = {(o, k): os.open(names[o][k], os.O_RDWR) for o in range(nout) for k in kinds}       # (read too: the writer maps the file)
    cursor = np.zeros((nout, 4), np.int64)                           # next free byte of every output file
    tot = {"blocks": 0, "reads": 0, "bases": 0, "stats": {}}
    out_bufs = out_bufs if out_bufs is not None else {}
    pins = None
    if pinned and not out_bufs and nblocks:
        # sized from the line index: the largest block's bytes (+ its mate's) bound every one of its outputs
        big = max((blocks[0][k][1] - blocks[0][k][0]) + ((blocks[1][k][1] - blocks[1][k][0]) if paired and k < len(blocks[1]) else 0)
                  for k in range(nblocks))
        pins, out_bufs = _pinned_outputs(kinds, big + 5 * 2 + 64)
    rounds = (nblocks + comm.world - 1) // comm.world
    for rd in range(rounds):
        k = rd * comm.world + comm.rank
        sizes = np.zeros((2, 4), np.int64)
        res = None
        if k < nblocks:
            b0, b1, _ = blocks[0][k]
            parts = [bufs[0][b0:b1]]
            if paired and k < len(blocks[1]):
                c0, c1, _ = blocks[1][k]
                parts.append(bufs[1][c0:c1])
            res = eng.fastq_job(parts, keep_headers=headers, fastq=want_fastq, streams=want_streams, hdr=want_hdr,
                                out=out_bufs)
            cut = {"fastq": res.part_fastq_off, "dna": res.part_stream_off, "qs": res.part_stream_off,
                   "hdr": res.part_hdr_off}
            raw = {"fastq": res.fastq, "dna": res.dna, "qs": res.qs, "hdr": res.hdr}
            blobs = {}
            for o in range(nout):
                for ki, kind in enumerate(KINDS):
                    if kind in kinds:
                        lo = cut[kind][o] if o < len(parts) else cut[kind][-1]
                        hi = cut[kind][o + 1] if o < len(parts) else cut[kind][-1]
                        sizes[o, ki] = hi - lo
                        if compress:                                 # this mate's share of the block as one container
                            blobs[(o, kind)] = eng.stream_compress(raw[kind][lo:hi])
                            sizes[o, ki] = len(blobs[(o, kind)])
            tot["blocks"] += 1; tot["reads"] += res.n_reads; tot["bases"] += res.total_bases
            for key, v in res.stats.items():
                if key.startswith("n_"):
                    continue
                tot["stats"][key] = tot["stats"].get(key, 0) + v
            if log:
                log(f"block {k + 1}/{nblocks}: {res.n_reads} reads, {res.total_bases} bases")
        allsz = comm.all_gather_i64(sizes.reshape(-1)).reshape(comm.world, 2, 4)
        before = allsz[:comm.rank].sum(axis=0)                       # blocks of this round that come first
        if res is not None:
            data = {"fastq": res.fastq, "dna": res.dna, "qs": res.qs, "hdr": res.hdr}
            for o in range(nout):
                for ki, kind in enumerate(KINDS):
                    if kind in kinds and sizes[o, ki]:
                        lo = cut[kind][o]
                        piece = blobs[(o, kind)] if compress else data[kind][lo:lo + int(sizes[o, ki])]
                        pwrite_all(fds[(o, kind)], piece, int(cursor[o, ki] + before[o, ki]), host)
        cursor += allsz.sum(axis=0)[:nout]
    for fd in fds.values():
        os.close(fd)
    if pins:
        for pb in pins.values():
            pb.free()
    comm.barrier()
    return tot


def deal_piles(counts, world):
    """The two-symbol piles (first 1..5, second 1..5; a second symbol 0 = suffixes of one base: never in a cluster) dealt to
    the ranks, largest first to the least loaded rank: [[(s, s2), ...] per rank] -- the same list on every rank."""
    piles = sorted(((int(counts[s][s2]), s, s2) for s in range(1, 6) for s2 in range(1, 6) if counts[s][s2]), reverse=True)
    load = [0] * world
    mine = [[] for _ in range(world)]
    for cnt, s, s2 in piles:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += cnt
        mine[r].append((s, s2))
    return mine


def run_global(eng, comm, inputs, names, headers=False, want_fastq=True, want_streams=False, want_hdr=False, log=None, out_bufs=None,
               pinned=False):
    """ONE collection over all ranks with the result of the unsharded run (k_global.hip): every rank parses its share of the
    file(s), the terminated text is exchanged, the two-symbol piles of the global eBWT are dealt to the ranks, the edits are
    combined with one all-reduce, and every rank writes its own reads.  Two input files (paired end) form ONE collection --
    all reads of file 1, then all reads of file 2, as `BFQzip_parallel.py -p -t 0` appends them -- and give two outputs."""
    import time
    import torch
    dev = torch.device(eng.tensor_device)
    host = eng.host
    tm = {}
    t_last = [time.perf_counter()]

    def lap(name):
        if dev.type == "cuda":
            torch.cuda.synchronize()
        now = time.perf_counter(); tm[name] = round(tm.get(name, 0.0) + now - t_last[0], 4); t_last[0] = now

    def tsync():
        """The engine runs on its own non-blocking stream (bfq_api.hip: hipStreamNonBlocking) and returns synchronised; what
        torch has queued on ITS stream (clones, copies, collectives) must be complete before the engine reads or overwrites
        those tensors -- stated here explicitly instead of resting on the timing helper."""
        if dev.type == "cuda":
            torch.cuda.current_stream().synchronize()
    nf = len(inputs)
    W, r = comm.world, comm.rank
    bufs = [map_file(p) for p in inputs]
    parts, tlen = [], 0
    for buf in bufs:                                                 # my share of every file: reads R r / W .. R (r + 1) / W
        idx = TextIndex(buf, comm, host.text_line_counts, host.text_nth_newline)
        if idx.num_lines % 4:
            raise ValueError("FASTQ: number of lines is not a multiple of 4")
        R = idx.num_lines // 4
        b0, b1 = idx.line_start(4 * (R * r // W)), idx.line_start(4 * (R * (r + 1) // W))
        parts.append(buf[b0:b1]); tlen += b1 - b0
    lap("index")
    Np, Tp = eng.glob_begin(parts)                                    # reads / bases of each of my parts
    lap("upload+parse")
    sz = comm.all_gather_i64(np.array([Np[f] + Tp[f] for f in range(nf)], np.int64))       # rows [rank][file]
    # global text = file 0's parts in rank order, then file 1's: base[f][k] = first row of rank k's part of file f
    flat = np.concatenate([[0], np.cumsum(sz.T.reshape(-1))]).astype(np.int64)
    base = flat[:-1].reshape(nf, W)
    n = int(flat[-1])
    # 64 bytes of padding: the pile kernels read the text 16 bytes at a time (k_piles.hip)
    t8 = torch.empty(n + 64, dtype=torch.uint8, device=dev); q8 = torch.empty_like(t8)
    t8[n:] = 0; q8[n:] = 0
    myrows = [int(sz[r][f]) for f in range(nf)]
    if sum(myrows):
        lt = torch.empty(sum(myrows), dtype=torch.uint8, device=dev); lq = torch.empty_like(lt)
        eng.glob_local_text(lt, lq)                                   # my parts back to back
        o = 0
        for f in range(nf):
            t8[int(base[f][r]):int(base[f][r]) + myrows[f]] = lt[o:o + myrows[f]]
            q8[int(base[f][r]):int(base[f][r]) + myrows[f]] = lq[o:o + myrows[f]]
            o += myrows[f]
        del lt, lq
    for f in range(nf):                                               # every rank ends up with the whole text
        for src in range(W):
            lo, hi = int(base[f][src]), int(base[f][src]) + int(sz[src][f])
            comm.broadcast_(t8[lo:hi], src)
            comm.broadcast_(q8[lo:hi], src)
    lap("text exchange")
    tot = {"blocks": 1, "reads": sum(Np), "bases": sum(Tp), "stats": {}, "seconds": tm}
    # The stream files' sizes are known from here on (one byte per row): they are created now, and this rank's byte range of
    # each is mapped and populated by a helper thread while the piles are sorted -- the streams then leave the GPU straight
    # into the files' pages, with nothing to write afterwards (one input file; with two the shares of both go through a buffer).
    kinds = [k for k, w in zip(KINDS, (want_fastq, want_streams, want_streams, want_hdr)) if w]
    if comm.rank == 0:
        for f in range(nf):
            for k in kinds:
                open(names[f][k], "wb").close()
    comm.barrier()
    direct, direct_thread = {}, None
    FileRange = getattr(host, "FileRange", None)
    if want_streams and nf == 1 and FileRange is not None and not out_bufs and not pinned and sum(myrows) >= (1 << 20):
        import threading
        off0 = int(sz[:r, 0].sum())

        def _map_streams():
            for kind in ("dna", "qs"):
                fd = os.open(names[0][kind], os.O_RDWR)
                fr = FileRange(fd, off0, myrows[0])
                os.close(fd)
                if fr.array is not None:
                    direct[kind] = fr
        direct_thread = threading.Thread(target=_map_streams)
        direct_thread.start()
    sym = torch.empty_like(t8); qual = torch.empty_like(t8)
    if n:
        tsync()
        counts = eng.glob_pile_counts(t8, n)
        mine = deal_piles(counts, W)[r]
        eng.glob_init_out(t8, q8, n, sym, qual)
        if W > 1:
            osym, oqual = sym.clone(), qual.clone()
            tsync()                                                  # the clones are read before the piles edit sym / qual
        for s, s2 in mine:
            st = eng.glob_run_pile(t8, q8, n, s, s2, sym, qual)
            for key, v in st.items():
                if not key.startswith("n_"):
                    tot["stats"][key] = tot["stats"].get(key, 0) + v
            if log:
                log(f"pile {'#ACGNT'[s]}{'#ACGNT'[s2]}: {st['n_rows']} rows, {st['num_clust']} clusters")
        del t8, q8
        lap("piles")
        segs = [(int(base[f][r]), int(base[f][r]) + myrows[f]) for f in range(nf)]
        if W > 1:
            sym ^= osym; qual ^= oqual                               # what this rank's piles changed (zero elsewhere)
            comm.all_reduce_sum_(sym); comm.all_reduce_sum_(qual)
            dna = torch.cat([osym[a:b] ^ sym[a:b] for a, b in segs]); qs = torch.cat([oqual[a:b] ^ qual[a:b] for a, b in segs])
            del osym, oqual
        elif nf == 1:                                                # one rank, one file: the edited streams are the result as they are
            dna, qs = sym[segs[0][0]:segs[0][1]], qual[segs[0][0]:segs[0][1]]
        else:
            dna = torch.cat([sym[a:b] for a, b in segs]); qs = torch.cat([qual[a:b] for a, b in segs])
        del sym, qual
        lap("delta all-reduce")
    else:
        dna = torch.empty(0, dtype=torch.uint8, device=dev); qs = dna.clone()
    tsync()
    pins = None
    if direct_thread is not None:
        direct_thread.join()
        if len(direct) == 2:
            out_bufs = {k: fr.array for k, fr in direct.items()}
        else:
            for fr in direct.values():
                fr.close()
            direct = {}
    if pinned and not out_bufs:
        pins, out_bufs = _pinned_outputs([k for k, w in zip(KINDS, (want_fastq, want_streams, want_streams, want_hdr)) if w], int(tlen) + 64)
    res = eng.glob_finish(dna, qs, keep_headers=headers, fastq=want_fastq, streams=want_streams, hdr=want_hdr, text_len=int(tlen), nparts=nf,
                          **({"out": out_bufs} if out_bufs else {}))
    # outputs at their final offsets: output f = the shares of part f of all ranks, in rank order
    data = {"fastq": res.fastq, "dna": res.dna, "qs": res.qs, "hdr": res.hdr}
    cut = {"fastq": res.part_fastq_off, "dna": res.part_stream_off, "qs": res.part_stream_off, "hdr": res.part_hdr_off}
    sizes = np.zeros((nf, 4), np.int64)
    for f in range(nf):
        for ki, kind in enumerate(KINDS):
            if kind in kinds:
                sizes[f, ki] = cut[kind][f + 1] - cut[kind][f]
    allsz = comm.all_gather_i64(sizes.reshape(-1)).reshape(W, nf, 4)
    before = allsz[:comm.rank].sum(axis=0)
    for f in range(nf):
        for ki, kind in enumerate(KINDS):
            if kind in kinds:
                if kind in direct:                                   # already in the file's pages
                    assert int(before[f, ki]) == direct[kind].offset and int(sizes[f, ki]) == direct[kind].nbytes
                    continue
                fd = os.open(names[f][kind], os.O_RDWR)
                if sizes[f, ki]:
                    pwrite_all(fd, data[kind][cut[kind][f]:cut[kind][f + 1]], int(before[f, ki]), host)
                os.close(fd)
    if pins or direct:
        del res, data
        out_bufs = None
    if pins:
        for pb in pins.values():
            pb.free()
    for fr in direct.values():
        fr.close()
    comm.barrier()
    lap("format+write")
    keys = sorted(tot["stats"]) if tot["stats"] else ["num_clust", "num_clust_discarded", "num_clust_amb_discarded", "num_clust_mod",
                                                     "num_clust_alleq", "bases_inside", "qs_smoothed", "modified"]
    allst = comm.all_gather_i64(np.array([tot["stats"].get(k, 0) for k in keys], np.int64))
    tot["stats_all_ranks"] = {k: int(v) for k, v in zip(keys, allst.sum(axis=0))}
    return tot


def main(argv=None):
    """`python -m torch.distributed.run --nproc-per-node G -m bfqzip_amd.parallel in.fastq [in2.fastq -p] -o OUT -t n [-H]`

    The multi-GPU counterpart of `BFQzip_parallel.py in.fastq [in2.fastq -p] -o OUT -t n -0` (same flags, same
    block split, same output names: OUT<ext> or OUT_1<ext> / OUT_2<ext>), one block per GPU at a time.  --m2 / --m3
    additionally write the streams BFQzip.py cuts with sed (<fastq>.dna, <fastq>.qs; --m3: OUT.h and header lines
    kept).  Without torchrun it runs all blocks on GPU 0."""
    import argparse
    import torch
    from . import api
    ap = argparse.ArgumentParser(prog="bfqzip_amd.parallel")
    ap.add_argument("input", nargs="+")
    ap.add_argument("-o", "--out", default="")
    ap.add_argument("-T", "--mcl", default="", help="minimum context length (bfq_int -k)")
    ap.add_argument("-Q", "--rv", default="", help="constant replacement value (bfq_int -v)")
    ap.add_argument("-H", "--headers", action="store_true", help="store original headers")
    ap.add_argument("-0", "--m0", action="store_true", help="do not compress (the default here; see --compress)")
    ap.add_argument("--compress", action="store_true",
                    help="step 5 on the GPU: every output goes through the stream codec, files get the suffix .bsc (one container per block)")
    ap.add_argument("-p", "--paired", action="store_true")
    ap.add_argument("-t", "--threads", type=int, default=0, help="number of blocks")
    ap.add_argument("-c", "--check", action="store_true", help="accepted: records are always checked on the GPU")
    ap.add_argument("-v", type=int, default=0)
    ap.add_argument("--m2", action="store_true", help="also write <fastq>.dna and <fastq>.qs (BFQzip.py:231-249)")
    ap.add_argument("--m3", action="store_true", help="--m2 plus OUT.h, headers kept (BFQzip.py:60-64,192-201)")
    ap.add_argument("--streams-only", action="store_true", help="with --m2/--m3: do not write the merged FASTQ text")
    ap.add_argument("--pinned", action="store_true", help="page-locked output buffers (direct DMA)")
    ap.add_argument("--global", dest="glob", action="store_true",
                    help="ONE eBWT over the whole input, its piles dealt to the GPUs: the result of the unsharded run (-t is ignored)")
    ap.add_argument("--M", type=int, default=2); ap.add_argument("--B", type=int, default=0)
    a = ap.parse_args(argv)
    if a.paired and len(a.input) != 2:
        print("=== ERROR ===\npaired end mode", file=sys.stderr)
        return 1
    if a.m3:
        a.headers = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    if torch.cuda.device_count():
        torch.cuda.set_device(local)                                 # tensors of the global mode and RCCL use the engine's GPU
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BFQ_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local) if (dist and dist.get_backend() == "nccl") else None
    comm = Comm(dist, dev)
    par = dict(m=5, M=a.M, B=a.B)                                    # -m 5: what BFQzip.py passes (BFQzip.py:215)
    if a.mcl:
        par["k"] = int(a.mcl)
    if a.rv:
        par["v"] = ord(a.rv)
    eng = api.Engine(local, **par)
    names = output_names(a.input, a.out, a.paired)
    streams = a.m2 or a.m3
    log = (lambda m: print(f"[rank {comm.rank}] {m}", flush=True)) if a.v else None
    if a.glob:
        tot = run_global(eng, comm, a.input[:2 if a.paired else 1], names, headers=a.headers, want_fastq=not (streams and a.streams_only),
                         want_streams=streams, want_hdr=a.m3, log=log, pinned=a.pinned)
    else:
        tot = run_files(eng, comm, a.input, a.threads, names, paired=a.paired, headers=a.headers,
                        want_fastq=not (streams and a.streams_only), want_streams=streams, want_hdr=a.m3, log=log,
                        compress=a.compress and not a.m0, pinned=a.pinned)
    if a.v:
        print(f"[rank {comm.rank}] {tot}", flush=True)
    eng.close()
    if dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
