"""FASTQ <-> (bases, quals, read offsets, headers) conversion, numpy only.

Host-side mirror of the only data handling the reference's drivers do around
the hot path: 4-line records in, 4-line records out (BFQzip.py:192-251 `sed -n
1~4p/2~4p/4~4p`; bfq_int.cpp:797-810 writes `@\\n` or the header line, bases,
`+\\n`, quals).
"""
import numpy as np


def parse_fastq_bytes(buf):
    """bytes/ndarray of a 4-line-record FASTQ -> (bases u8, quals u8, roff u64[N+1], headers list[bytes])."""
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    if a.size == 0:
        return (np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint64), [])
    nl = np.flatnonzero(a == 10)
    if a[-1] != 10:                      # last line without newline
        nl = np.append(nl, a.size)
    if nl.size % 4 != 0:
        raise ValueError("FASTQ: number of lines is not a multiple of 4")
    starts = np.empty(nl.size, np.int64)
    starts[0] = 0
    starts[1:] = nl[:-1] + 1
    n = nl.size // 4
    s_seq, e_seq = starts[1::4], nl[1::4]
    s_q, e_q = starts[3::4], nl[3::4]
    lens = e_seq - s_seq
    if np.any(lens != (e_q - s_q)):
        raise ValueError("FASTQ: len(DNA) != len(QS) in some record")   # checkFASTQ.py:18-32
    roff = np.zeros(n + 1, np.uint64)
    roff[1:] = np.cumsum(lens)
    total = int(roff[-1])
    # gather via a repeat/arange index (vectorised)
    idx = np.repeat(s_seq - roff[:-1].astype(np.int64), lens) + np.arange(total, dtype=np.int64)
    bases = a[idx]
    idxq = np.repeat(s_q - roff[:-1].astype(np.int64), lens) + np.arange(total, dtype=np.int64)
    quals = a[idxq]
    hs, he = starts[0::4], nl[0::4]
    headers = [a[hs[i]:he[i]].tobytes() for i in range(n)] if n <= 2_000_000 else None
    return bases, quals, roff, headers


def read_fastq(path):
    with open(path, "rb") as f:
        return parse_fastq_bytes(f.read())


def format_fastq(bases, quals, roff, headers=None):
    """Inverse of parse: bfq_int.cpp:797-810 record layout. headers: list of bytes lines without newline, or None -> '@'."""
    n = len(roff) - 1
    out = bytearray()
    bb = bases.tobytes()
    qq = quals.tobytes()
    for i in range(n):
        s, e = int(roff[i]), int(roff[i + 1])
        out += (headers[i] if headers is not None else b"@") + b"\n"
        out += bb[s:e] + b"\n+\n" + qq[s:e] + b"\n"
    return bytes(out)


def format_lines(data, roff):
    """One line per read (`sed -n 2~4p` / `4~4p` of the FASTQ, BFQzip.py:20-21): the streams OUT.fq.dna / OUT.fq.qs."""
    roff = np.asarray(roff, np.int64)
    n = len(roff) - 1
    out = np.full(int(roff[-1]) + n, 10, np.uint8)
    if len(data):
        lens = np.diff(roff)
        dst = np.arange(int(roff[-1]), dtype=np.int64) + np.repeat(np.arange(n, dtype=np.int64), lens)
        out[dst] = np.asarray(data, np.uint8)[:int(roff[-1])]
    return out.tobytes()
