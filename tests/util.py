import hashlib, json, os
import numpy as np
from bfqzip_amd import fastq

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_index():
    return json.load(open(os.path.join(GOLDEN, "index.json")))


def golden_set(name):
    b, q, r, h = fastq.read_fastq(os.path.join(GOLDEN, name + ".fastq"))
    bwt = np.fromfile(os.path.join(GOLDEN, name + ".bwt"), np.uint8)
    qs = np.fromfile(os.path.join(GOLDEN, name + ".bwt.qs"), np.uint8)
    lcp = np.fromfile(os.path.join(GOLDEN, name + ".lcp16"), np.uint16)
    return b, q, r, h, bwt, qs, lcp


def parse_case(key):
    """'M2B0 -m 5 -k 8 ...' -> dict of engine/oracle parameters (+ 'H' for headers)."""
    toks = key.split()
    M, B = int(toks[0][1]), int(toks[0][3])
    d = dict(M=M, B=B, k=16, m=2, v=ord(">"), f=40, t=20)
    i, hdr = 1, False
    while i < len(toks):
        if toks[i] == "-H":
            hdr = True; i += 1; continue
        d[toks[i][1]] = int(toks[i + 1]); i += 2
    return d, hdr


def md5(b):
    return hashlib.md5(b).hexdigest()


def random_reads(rng, nreads, lmin, lmax, glen=None, p_n=0.05, dup=0.2, qlo=33, qhi=74):
    """Random small read sets: variable length, duplicates, N's, full quality range."""
    glen = glen or max(lmax * 2, nreads * (lmin + lmax) // 2 // 8 + lmax + 1)
    genome = rng.integers(0, 4, glen)
    reads, quals = [], []
    for _ in range(nreads):
        if reads and rng.random() < dup:
            k = int(rng.integers(0, len(reads)))
            s = reads[k].copy()
        else:
            L = int(rng.integers(lmin, lmax + 1))
            st = int(rng.integers(0, glen - L + 1))
            s = genome[st:st + L].copy()
            err = rng.random(L) < 0.03
            s[err] = (s[err] + rng.integers(1, 4, int(err.sum()))) % 4
        s = np.array(list(b"ACGT"), np.uint8)[s] if s.dtype != np.uint8 else s
        nmask = rng.random(len(s)) < p_n
        s = s.copy(); s[nmask] = ord("N")
        reads.append(s)
        quals.append(rng.integers(qlo, qhi + 1, len(s)).astype(np.uint8))
    roff = np.zeros(nreads + 1, np.uint64)
    roff[1:] = np.cumsum([len(s) for s in reads])
    cat = lambda xs: np.concatenate(xs) if sum(len(x) for x in xs) else np.zeros(0, np.uint8)
    return cat(reads).astype(np.uint8), cat(quals).astype(np.uint8), roff


class OracleEngine:
    """Engine.fastq_job on the CPU oracle: drives bfqzip_amd.parallel in the CPU (gloo) tests, and is the checker the
    GPU engine's block outputs are compared with."""

    def __init__(self, orc, **params):
        from bfqzip_amd import api
        self.orc, self.p = orc, orc.params(**params)
        self.host = api.HostText

    def fastq_job(self, parts, keep_headers=False, fastq=True, streams=False, hdr=False, out=None):
        from bfqzip_amd import api, fastq as fqm
        texts = []
        for p in parts:
            b = bytes(np.asarray(p, np.uint8).tobytes()) if not isinstance(p, (bytes, bytearray)) else bytes(p)
            if b and not b.endswith(b"\n"):
                b += b"\n"
            texts.append(b)
        whole = b"".join(texts)
        b, q, r, h = fqm.parse_fastq_bytes(whole)
        ob, oq, st = self.orc.run_reads(b, q, r, self.p)
        res = api.JobResult()
        res.n_reads, res.total_bases = len(r) - 1, int(r[-1])
        res.fastq = np.frombuffer(fqm.format_fastq(ob, oq, r, h if keep_headers else None), np.uint8) if fastq else None
        res.dna = np.frombuffer(fqm.format_lines(ob, r), np.uint8) if streams else None
        res.qs = np.frombuffer(fqm.format_lines(oq, r), np.uint8) if streams else None
        res.hdr = np.frombuffer(fqm.format_headers(h), np.uint8) if hdr else None
        pr = [0]
        for t in texts:
            pr.append(pr[-1] + t.count(b"\n") // 4)
        hl = h.lengths() if len(h) else np.zeros(0, np.int64)
        L = np.diff(r.astype(np.int64))
        fsz = np.concatenate([[0], np.cumsum((hl if keep_headers else np.ones(len(L), np.int64)) + 2 * L + 5)])
        hsz = np.concatenate([[0], np.cumsum(hl + 1)])
        res.part_reads = pr
        res.part_fastq_off = [int(fsz[i]) for i in pr] if fastq else [0] * len(pr)
        res.part_stream_off = [int(r[i]) + i for i in pr] if streams else [0] * len(pr)
        res.part_hdr_off = [int(hsz[i]) for i in pr] if hdr else [0] * len(pr)
        res.stats = dict(st)
        return res


def decode_rows(bwt, term=ord("#")):
    """For every row of an eBWT: (read index, suffix start) found by the LF walks, plus the reads (small inputs only).
    Returns (suffix strings per row as bytes, reads as list of bytes)."""
    bwt = np.asarray(bwt, np.uint8)
    n = len(bwt)
    order = [term] + [ord(c) for c in "ACGNT"]
    F, acc = {}, 0
    for c in order:
        F[c] = acc; acc += int(np.count_nonzero(bwt == c))
    occ = {c: 0 for c in order}
    LF = np.zeros(n, np.int64)
    for r in range(n):
        c = int(bwt[r]); LF[r] = F[c] + occ[c]; occ[c] += 1
    N = F[ord("A")]
    suf = [None] * n
    reads = []
    for i in range(N):
        r, s = i, b""
        while True:
            suf[r] = s
            if bwt[r] == term:
                break
            s = bytes([bwt[r]]) + s; r = int(LF[r])
        reads.append(s)
    return suf, reads


def shuffle_ties(bwt, qs, rng):
    """Permute the rows of every block of identical suffixes independently: an eBWT of the same kind (one terminator
    symbol, decodable) whose ties are in no consistent order -- what another step-1 tool may produce."""
    suf, _ = decode_rows(bwt)
    bwt, qs = np.array(bwt, np.uint8), np.array(qs, np.uint8)
    j, n = 0, len(bwt)
    while j < n:
        e = j
        while e < n and suf[e] == suf[j]:
            e += 1
        if e - j > 1:
            p = rng.permutation(e - j) + j
            bwt[j:e] = bwt[p]; qs[j:e] = qs[p]
        j = e
    return bwt, qs


def lcp_of_rows(suf):
    """LCP array by the convention of bfq_int.cpp:139-145 / bfq_ext.cpp:377-392 from decoded suffixes (terminators never match)."""
    n = len(suf)
    L = np.zeros(n, np.uint32)
    for r in range(1, n):
        a, b = suf[r - 1], suf[r]
        k = 0
        while k < len(a) and k < len(b) and a[k] == b[k]:
            k += 1
        L[r] = k
    return L
