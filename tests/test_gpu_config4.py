"""BASELINE.json configs[4] at its PER-GPU size: paired-end 2 x 150 M x 150 bp over 8 GPUs = one block of 18.75 M pairs per
GPU = 37.5 M reads, 5.66 G eBWT rows, M = 1 (mean-error smoothing), --m3 (DNA / QS / header streams), through
bfq_fastq_run_job with the two mate blocks as two parts (BFQzip_parallel.py:325-360 appends mate block k of file 2 to block
k of file 1; :153-172 cuts the output back by line count).  No oracle runs at this size: the checks are size-independent
properties, each over ALL the bytes (on the GPU, chunk by chunk), plus a 200 k-read block of the same generator against
the oracle."""
import numpy as np
import pytest
from bfqzip_amd import api

pytestmark = pytest.mark.gpu


def _lines_of(torch, dev, text, which, chunk=1 << 28):
    """Bytes of lines `which` (0..3) of every 4-line record of a FASTQ text (host uint8 array), newlines included,
    as a list of device tensors: chunk by chunk, the line number of a byte = newlines before it."""
    out, carry = [], 0
    for o in range(0, len(text), chunk):
        t = torch.from_numpy(np.asarray(text[o:o + chunk])).to(dev)
        nl = (t == 10)
        lid = torch.cumsum(nl, 0, dtype=torch.int32) - nl.to(torch.int32) + carry       # newlines strictly before the byte
        out.append(t[(lid & 3) == which])
        carry = (carry + int(nl.sum().item())) & 3
        del t, nl, lid
    return out


def _same(torch, pieces, ref_host, dev, chunk=1 << 28):
    """The concatenation of the device pieces equals the host array `ref_host`."""
    pos = 0
    for p in pieces:
        n = int(p.numel())
        for o in range(0, n, chunk):
            m = min(chunk, n - o)
            if not torch.equal(p[o:o + m], torch.from_numpy(np.asarray(ref_host[pos + o:pos + o + m])).to(dev)):
                return False
        pos += n
    return pos == len(ref_host)


def _count_diff(torch, a_pieces, b_host, dev, chunk=1 << 28):
    pos, k = 0, 0
    for p in a_pieces:
        n = int(p.numel())
        for o in range(0, n, chunk):
            m = min(chunk, n - o)
            k += int((p[o:o + m] != torch.from_numpy(np.asarray(b_host[pos + o:pos + o + m])).to(dev)).sum().item())
        pos += n
    assert pos == len(b_host)
    return k


def test_config4_share_two_parts_m1_streams_and_fastq():
    torch = pytest.importorskip("torch")
    free, total = torch.cuda.mem_get_info()
    if free < 230 * 2**30:
        pytest.skip(f"configs[4]'s per-GPU share needs 230 GiB of free HBM, {free / 2**30:.0f} GiB free")
    dev = torch.device("cuda:0")
    P, L = 18_750_000, 150                                            # reads per mate block
    eng = api.Engine(0, k=16, m=5, M=1, B=0)
    pins = []
    try:
        parts = []
        for f in range(2):                                            # mate block of file 1, mate block of file 2: one collection
            pb = api.PinnedBuffer(P * (2 * L + 30) + 4096); pins.append(pb)
            ln = eng.synth_fastq(api.synth_spec(P, L, seed=777, first=f * P, collection=2 * P), pb.array)
            parts.append(pb.array[:ln])
        tlen = sum(len(p) for p in parts)
        n = 2 * P * (L + 1)
        outs = {"fastq": api.PinnedBuffer(tlen + 64), "dna": api.PinnedBuffer(n + 64), "qs": api.PinnedBuffer(n + 64), "hdr": api.PinnedBuffer(2 * P * 24)}
        pins += list(outs.values())
        ob = {k: v.array for k, v in outs.items()}
        # 1. K above every LCP: no cluster, the FASTQ text of each part comes back byte for byte (sort + LF round trip at 5.66 G rows)
        eng.set_params(k=10000, m=5, M=1, B=0)
        r = eng.fastq_job(parts, keep_headers=True, fastq=True, streams=True, hdr=True, out=ob)
        assert r.stats["num_clust"] == 0 and r.n_reads == 2 * P and r.total_bases == 2 * P * L and r.stats["n_rows"] == n
        assert list(r.part_reads) == [0, P, 2 * P]
        assert list(r.part_fastq_off) == [0, len(parts[0]), tlen] and len(r.fastq) == tlen
        assert list(r.part_stream_off) == [0, P * (L + 1), n]
        for f in range(2):
            a, b = int(r.part_fastq_off[f]), int(r.part_fastq_off[f + 1])
            assert _same(torch, [torch.from_numpy(np.asarray(r.fastq[a:b][o:o + (1 << 28)])).to(dev) for o in range(0, b - a, 1 << 28)], parts[f], dev)
        # the streams of the inputs (= of this identity run): what the edited run is compared with
        in_dna = np.array(r.dna, copy=True); in_qs = np.array(r.qs, copy=True); in_hdr = np.array(r.hdr, copy=True)
        for f in range(2):                                            # stream = lines 2 / 4, header stream = lines 1 of the part
            a, b = int(r.part_stream_off[f]), int(r.part_stream_off[f + 1])
            assert _same(torch, _lines_of(torch, dev, parts[f], 1), in_dna[a:b], dev)
            assert _same(torch, _lines_of(torch, dev, parts[f], 3), in_qs[a:b], dev)
            ha, hb = int(r.part_hdr_off[f]), int(r.part_hdr_off[f + 1])
            assert _same(torch, _lines_of(torch, dev, parts[f], 0), in_hdr[ha:hb], dev)
        assert int(r.part_hdr_off[2]) == len(in_hdr)
        # 2. the real run: M = 1, K = 16, headers kept, FASTQ + the three streams in one pass
        eng.set_params(k=16, m=5, M=1, B=0)
        r = eng.fastq_job(parts, keep_headers=True, fastq=True, streams=True, hdr=True, out=ob)
        st = r.stats
        assert st["num_clust"] > 0 and st["modified"] > 0 and st["qs_smoothed"] > 0 and st["n_rows"] == n
        assert list(r.part_reads) == [0, P, 2 * P] and list(r.part_stream_off) == [0, P * (L + 1), n]
        assert list(r.part_fastq_off) == [0, len(parts[0]), tlen]     # lengths do not change: the OUT_1 / OUT_2 cut (BFQzip_parallel.py:153-172)
        assert np.array_equal(np.asarray(r.hdr), in_hdr)               # headers pass through
        dna_out = _lines_of(torch, dev, r.fastq, 1)                    # streams = lines 2 / 4 of the FASTQ written in the same pass
        assert _same(torch, dna_out, r.dna, dev)
        qs_out = _lines_of(torch, dev, r.fastq, 3)
        assert _same(torch, qs_out, r.qs, dev)
        assert _same(torch, _lines_of(torch, dev, r.fastq, 0), in_hdr, dev)
        plus = _lines_of(torch, dev, r.fastq, 2)
        assert sum(int(p.numel()) for p in plus) == 2 * (2 * P) and all(bool(((p == 43) | (p == 10)).all()) for p in plus)   # "+\n"
        # edits consistent with the statistics: every replacement changes a base; smoothing changes at most qs_smoothed qualities
        assert _count_diff(torch, dna_out, in_dna, dev) == st["modified"]
        dq = _count_diff(torch, qs_out, in_qs, dev)
        assert 0 < dq <= st["qs_smoothed"]
        del dna_out, qs_out, plus
        r2 = eng.fastq_job(parts, keep_headers=True, fastq=False, streams=True, hdr=False, out=ob)
        assert r2.stats == st                                          # deterministic
    finally:
        eng.close()
        for p in pins:
            p.free()
        torch.cuda.empty_cache()


def test_config4_block_against_the_oracle(orc):
    """One block of the same shape at 200 k reads (2 parts, M = 1, headers): FASTQ text and streams = the oracle's reads."""
    from bfqzip_amd import fastq
    P, L = 100_000, 150
    eng = api.Engine(0, k=16, m=5, M=1, B=0)
    try:
        parts = []
        for f in range(2):
            buf = np.zeros(P * (2 * L + 30) + 4096, np.uint8)
            ln = eng.synth_fastq(api.synth_spec(P, L, seed=777, first=f * P, collection=2 * P), buf)
            parts.append(buf[:ln])
        r = eng.fastq_job(parts, keep_headers=True, fastq=True, streams=True, hdr=True)
        b, q, roff, hdrs = fastq.parse_fastq_bytes(b"".join(p.tobytes() for p in parts))
        ob, oq, ost = orc.run_reads(b, q, roff, orc.params(K=16, m=5, M=1, B=0))
        assert np.asarray(r.fastq).tobytes() == fastq.format_fastq(ob, oq, roff, hdrs)
        assert np.asarray(r.dna).tobytes() == fastq.format_lines(ob, roff) and np.asarray(r.qs).tobytes() == fastq.format_lines(oq, roff)
        for k in ost:
            assert ost[k] == r.stats[k], k
    finally:
        eng.close()
