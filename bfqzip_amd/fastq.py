"""FASTQ <-> (bases, quals, read offsets, headers) conversion, numpy only (vectorised: no per-read Python loops).

Host-side mirror of the only data handling the reference's drivers do around
the hot path: 4-line records in, 4-line records out (BFQzip.py:192-251 `sed -n
1~4p/2~4p/4~4p`; bfq_int.cpp:797-810 writes `@\\n` or the header line, bases,
`+\\n`, quals).  Used by the tests, the oracle-backed engine of the CPU tests and
bench.py's CPU baseline; the product path parses and formats on the GPU (k_fastq.hip).
"""
import numpy as np


class HeaderSpans:
    """The header lines of a FASTQ text as spans into its buffer (any number of reads)."""

    def __init__(self, buf, starts, ends):
        self.buf, self.starts, self.ends = buf, np.asarray(starts, np.int64), np.asarray(ends, np.int64)

    def __len__(self):
        return len(self.starts)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return HeaderSpans(self.buf, self.starts[i], self.ends[i])
        return self.buf[self.starts[i]:self.ends[i]].tobytes()

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __eq__(self, other):
        return list(self) == list(other)

    def lengths(self):
        return self.ends - self.starts

    def concat(self):
        """All header bytes back to back (no newlines)."""
        return _gather_segments(self.buf, self.starts, self.lengths())

    @staticmethod
    def from_list(lines):
        lens = np.array([len(x) for x in lines], np.int64)
        ends = np.cumsum(lens)
        return HeaderSpans(np.frombuffer(b"".join(lines), np.uint8), ends - lens, ends)


def _seg_index(starts, lens):
    """Index array visiting [starts[i], starts[i]+lens[i]) for every i in order."""
    lens = np.asarray(lens, np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, np.int64)
    ex = np.cumsum(lens) - lens
    return np.repeat(np.asarray(starts, np.int64) - ex, lens) + np.arange(total, dtype=np.int64)


def _gather_segments(buf, starts, lens):
    return np.asarray(buf)[_seg_index(starts, lens)]


def parse_fastq_bytes(buf):
    """bytes/ndarray of a 4-line-record FASTQ -> (bases u8, quals u8, roff u64[N+1], headers HeaderSpans)."""
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    if a.size == 0:
        return (np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint64), HeaderSpans(a, [], []))
    nl = np.flatnonzero(a == 10)
    if a[-1] != 10:                      # last line without newline
        nl = np.append(nl, a.size)
    if nl.size % 4 != 0:
        raise ValueError("FASTQ: number of lines is not a multiple of 4")
    starts = np.empty(nl.size, np.int64)
    starts[0] = 0
    starts[1:] = nl[:-1] + 1
    ends = nl.astype(np.int64).copy()
    # CR before LF is dropped from lines 2 and 4 (header lines pass through verbatim), as k_fq_records does
    for k in (1, 3):
        e = ends[k::4]
        has = (e > starts[k::4]) & (a[np.maximum(e - 1, 0)] == 13)
        ends[k::4] = e - has
    n = nl.size // 4
    s_seq, e_seq = starts[1::4], ends[1::4]
    s_q, e_q = starts[3::4], ends[3::4]
    lens = e_seq - s_seq
    if np.any(lens != (e_q - s_q)):
        raise ValueError("FASTQ: len(DNA) != len(QS) in some record")   # checkFASTQ.py:18-32
    roff = np.zeros(n + 1, np.uint64)
    roff[1:] = np.cumsum(lens)
    bases = a[_seg_index(s_seq, lens)]
    quals = a[_seg_index(s_q, lens)]
    return bases, quals, roff, HeaderSpans(a, starts[0::4], nl[0::4])


def read_fastq(path):
    with open(path, "rb") as f:
        return parse_fastq_bytes(f.read())


def format_fastq(bases, quals, roff, headers=None):
    """Inverse of parse: bfq_int.cpp:797-810 record layout. headers: HeaderSpans / list of bytes lines without
    newline, or None -> '@'."""
    roff = np.asarray(roff, np.int64)
    n = len(roff) - 1
    if n <= 0:
        return b""
    L = np.diff(roff)
    if headers is not None and not isinstance(headers, HeaderSpans):
        headers = HeaderSpans.from_list(list(headers))
    hl = headers.lengths() if headers is not None else np.ones(n, np.int64)
    size = hl + 2 * L + 5
    off = np.cumsum(size) - size
    out = np.empty(int(size.sum()), np.uint8)
    if headers is not None:
        out[_seg_index(off, hl)] = headers.concat()
    else:
        out[off] = ord("@")
    o = off + hl
    out[o] = 10
    out[_seg_index(o + 1, L)] = np.asarray(bases, np.uint8)[:int(roff[-1])]
    out[o + 1 + L] = 10
    out[o + 2 + L] = ord("+")
    out[o + 3 + L] = 10
    out[_seg_index(o + 4 + L, L)] = np.asarray(quals, np.uint8)[:int(roff[-1])]
    out[o + 4 + 2 * L] = 10
    return out.tobytes()


def format_lines(data, roff):
    """One line per read (`sed -n 2~4p` / `4~4p` of the FASTQ, BFQzip.py:20-21): the streams OUT.fq.dna / OUT.fq.qs."""
    roff = np.asarray(roff, np.int64)
    n = len(roff) - 1
    out = np.full(int(roff[-1]) + n, 10, np.uint8)
    if len(data):
        lens = np.diff(roff)
        out[_seg_index(roff[:-1] + np.arange(n, dtype=np.int64), lens)] = np.asarray(data, np.uint8)[:int(roff[-1])]
    return out.tobytes()


def format_headers(headers):
    """`sed -n 1~4p` of the input (BFQzip.py:19): the stream OUT.h."""
    if not isinstance(headers, HeaderSpans):
        headers = HeaderSpans.from_list(list(headers))
    n = len(headers)
    hl = headers.lengths()
    off = np.cumsum(hl + 1) - (hl + 1)
    out = np.full(int((hl + 1).sum()), 10, np.uint8)
    if n:
        out[_seg_index(off, hl)] = headers.concat()
    return out.tobytes()
