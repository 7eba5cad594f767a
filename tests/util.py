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

    def stream_compress(self, data, out=None):
        return self.orc.codec_encode(np.ascontiguousarray(data, np.uint8))

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


class OracleGlobalEngine(OracleEngine):
    """The per-GPU pieces of the global mode (bfqzip_amd.api.Engine.glob_*) on the CPU oracle, torch CPU tensors: drives
    bfqzip_amd.parallel.run_global in the CPU (gloo) tests.  A pile's edits = the unsharded oracle result at the positions
    whose following suffix starts with the pile's two symbols -- the contract of bfq_glob_run_pile."""
    tensor_device = "cpu"
    CODE = {ord("A"): 1, ord("C"): 2, ord("G"): 3, ord("N"): 4, ord("T"): 5}

    def __init__(self, orc, **params):
        super().__init__(orc, **params)
        self.B = params.get("B", 0)
        self._full = None

    def glob_begin(self, parts):
        from bfqzip_amd import fastq as fqm
        texts = []
        for p in parts:
            t = bytes(np.asarray(p, np.uint8).tobytes())
            if t and not t.endswith(b"\n"):
                t += b"\n"
            texts.append(t)
        self.text = b"".join(texts)
        b, q, r, h = fqm.parse_fastq_bytes(self.text)
        self.blk = (b, q, r, h)
        self.pr = [0]
        for t in texts:
            self.pr.append(self.pr[-1] + t.count(b"\n") // 4)
        return ([self.pr[i + 1] - self.pr[i] for i in range(len(texts))],
                [int(r[self.pr[i + 1]]) - int(r[self.pr[i]]) for i in range(len(texts))])

    def glob_local_text(self, t8, q8):
        b, q, r, h = self.blk
        n = len(b) + len(r) - 1
        T = np.zeros(n, np.uint8); Q = np.full(n, ord("#"), np.uint8)
        L = np.diff(r.astype(np.int64))
        idx = fqm_seg(r[:-1].astype(np.int64) + np.arange(len(L)), L)
        lut = np.zeros(256, np.uint8)
        for k, v in self.CODE.items():
            lut[k] = v
        T[idx] = lut[b]; Q[idx] = q
        t8.numpy()[:] = T; q8.numpy()[:] = Q

    def glob_pile_counts(self, t8, n):
        T = t8.numpy()[:n].astype(np.int64)
        cnt = np.zeros((6, 6), np.uint64)
        cnt[0][0] = int(np.count_nonzero(T == 0))
        nxt = np.concatenate([T[1:], [0]])
        for s in range(1, 6):
            m = T == s
            for s2 in range(6):
                cnt[s][s2] = int(np.count_nonzero(m & (nxt == s2)))
        return cnt

    def glob_init_out(self, t8, q8, n, sym, qual):
        T = t8.numpy()[:n]
        lut = np.frombuffer(b"\nACGNT\n\n", np.uint8)
        sym.numpy()[:n] = lut[T]
        qual.numpy()[:n] = np.where(T == 0, 10, q8.numpy()[:n])

    def _global_result(self, t8, q8, n):
        if self._full is None:
            T, Q = t8.numpy()[:n], q8.numpy()[:n]
            ends = np.flatnonzero(T == 0)
            lens = np.diff(np.concatenate([[-1], ends])) - 1
            roff = np.zeros(len(ends) + 1, np.uint64); roff[1:] = np.cumsum(lens)
            keep = T != 0
            lut = np.frombuffer(b"#ACGNT##", np.uint8)
            b, q = lut[T[keep]], Q[keep]
            p0 = self.orc.params(K=self.p.K, m=self.p.m, v=self.p.v, f=self.p.f, t=self.p.t, M=self.p.M, B=0)
            ob, oq, st = self.orc.run_reads(b, q, roff, p0)
            fs = np.full(n, 10, np.uint8); fq = np.full(n, 10, np.uint8)
            fs[keep] = ob; fq[keep] = oq
            self._full = (fs, fq, st)
        return self._full

    def glob_run_pile(self, t8, q8, n, s, s2, sym, qual):
        fs, fq, st = self._global_result(t8, q8, n)
        T = t8.numpy()[:n]
        nxt = np.concatenate([T[1:], [0]]); nxt2 = np.concatenate([T[2:], [0, 0]])
        m = (T != 0) & (nxt == s) & (nxt2 == s2)
        sym.numpy()[:n][m] = fs[m]; qual.numpy()[:n][m] = fq[m]
        cnt = self.glob_pile_counts(t8, n)
        first = min((a, b) for a in range(1, 6) for b in range(1, 6) if cnt[a][b])
        out = dict(st) if (s, s2) == first else {k: 0 for k in st}
        out["n_rows"] = int(cnt[s][s2])
        return out

    def glob_finish(self, dna, qs, keep_headers=False, fastq=True, streams=False, hdr=False, text_len=0, nparts=1):
        from bfqzip_amd import api, fastq as fqm
        b, q, r, h = self.blk
        d, s = dna.numpy().copy(), qs.numpy().copy()
        if self.B:
            lut = np.arange(256, dtype=np.uint8)
            for c in range(256):
                v = (c - 256 if c > 127 else c) - 33
                for lo, w in ((40, 40), (35, 37), (30, 33), (25, 27), (20, 22), (10, 15), (2, 6)):
                    if v >= lo:
                        v = w
                        break
                lut[c] = (v + 33) & 0xFF
            lut[10] = 10
            s = lut[s]
        keep = d != 10
        ob, oq = d[keep], s[keep]
        res = api.JobResult()
        res.n_reads, res.total_bases = len(r) - 1, int(r[-1])
        res.fastq = np.frombuffer(fqm.format_fastq(ob, oq, r, h if keep_headers else None), np.uint8) if fastq else None
        res.dna = d if streams else None
        res.qs = s if streams else None
        res.hdr = np.frombuffer(fqm.format_headers(h), np.uint8) if hdr else None
        hl = h.lengths() if len(h) else np.zeros(0, np.int64)
        Ls = np.diff(r.astype(np.int64))
        fsz = np.concatenate([[0], np.cumsum((hl if keep_headers else np.ones(len(Ls), np.int64)) + 2 * Ls + 5)])
        hsz = np.concatenate([[0], np.cumsum(hl + 1)])
        res.part_reads = list(self.pr)
        res.part_fastq_off = [int(fsz[i]) for i in self.pr]
        res.part_stream_off = [int(r[i]) + i for i in self.pr]
        res.part_hdr_off = [int(hsz[i]) for i in self.pr]
        res.stats = {}
        return res


def fqm_seg(starts, lens):
    from bfqzip_amd import fastq as fqm
    return fqm._seg_index(starts, lens)
