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
