"""Executable model of k_bfs.hip's interval refinement (base-string intervals + leaf blocks of identical suffixes):
the algorithm of the GPU kernel in a few lines of Python, checked against LCPs computed directly from the decoded
suffixes, incl. eBWTs whose identical suffixes are in an inconsistent order (what bfq_int accepts, one terminator
symbol: bfq_int.cpp:139-145,183-300).  Test infrastructure (tests/test_bfs_model.py); also runnable: python tests/bfs_model.py [seed]."""
import numpy as np, sys

def ebwt(reads, rng=None, shuffle_ties=False):
    rows = []
    for i, r in enumerate(reads):
        for k in range(len(r) + 1):
            rows.append((r[k:], i, k))
    # order: suffix string with terminator smaller than bases; ties by read index (or shuffled)
    key = lambda t: ([ "ACGNT".index(c) + 1 for c in t[0]] + [0], t[1])
    rows.sort(key=key)
    if shuffle_ties:
        # permute rows inside blocks of identical suffix strings, independently per block (inconsistent order)
        out = []; j = 0
        while j < len(rows):
            e = j
            while e < len(rows) and rows[e][0] == rows[j][0]: e += 1
            blk = rows[j:e]
            perm = rng.permutation(len(blk))
            out += [blk[p] for p in perm]
            j = e
        rows = out
    bwt = [reads[i][k - 1] if k else "#" for (_, i, k) in rows]
    return bwt

def decode(bwt):
    n = len(bwt); N = bwt.count("#")
    order = "#ACGNT"
    cnt = {c: bwt.count(c) for c in order}
    F = {}; acc = 0
    for c in order: F[c] = acc; acc += cnt[c]
    occ = {c: 0 for c in order}; LF = [0] * n
    for r in range(n):
        c = bwt[r]; LF[r] = F[c] + occ[c]; occ[c] += 1
    suf = [None] * n
    for i in range(N):
        r = i; s = ""
        while True:
            if suf[r] is not None: return None, None    # not a path structure
            suf[r] = s
            if bwt[r] == "#": break
            s = bwt[r] + s; r = LF[r]
    if any(x is None for x in suf): return None, None
    return suf, F

def lcp_direct(suf):
    n = len(suf); L = [0] * n
    for r in range(1, n):
        a, b = suf[r - 1], suf[r]; k = 0
        while k < len(a) and k < len(b) and a[k] == b[k]: k += 1
        L[r] = k
    return L

def bfs(bwt):
    n = len(bwt); N = bwt.count("#")
    order = "ACGNT"
    tot = {c: bwt.count(c) for c in order}
    F = {}; acc = N
    for c in order: F[c] = acc; acc += tot[c]
    pre = {c: [0] * (n + 1) for c in order}
    for r in range(n):
        for c in order: pre[c][r + 1] = pre[c][r] + (bwt[r] == c)
    UNSET = -1
    lcp = [UNSET] * (n + 1)
    lcp[0] = 0; lcp[n] = 0
    q = []
    # level 0 children: leaf block of all terminator suffixes, one interval per base
    if N:
        for p in range(1, N): lcp[p] = 0
        lcp[N] = 0
        q.append((0, N - 1, True))
    for c in order:
        if tot[c]:
            lcp[F[c] + tot[c]] = 0
            q.append((F[c], F[c] + tot[c] - 1, False))
    level = 1; work = 0
    while q:
        nq = []
        for lb, rb, leaf in q:
            work += 1
            for c in order:
                a, b = pre[c][lb], pre[c][rb + 1]
                if b > a:
                    nlb, nrb = F[c] + a, F[c] + b - 1
                    if leaf:
                        for p in range(nlb + 1, nrb + 1):
                            assert lcp[p] == UNSET, "internal slot already set"
                            lcp[p] = level
                        claim = lcp[nrb + 1] == UNSET
                        if claim: lcp[nrb + 1] = level
                        if claim or nrb > nlb: nq.append((nlb, nrb, True))
                    elif lcp[nrb + 1] == UNSET:
                        lcp[nrb + 1] = level
                        nq.append((nlb, nrb, False))
        q = nq; level += 1
    return lcp[:n], work


def check(seed, iters):
    """Random small collections (duplicates, prefixes, N, empty reads), each as built and with shuffled ties.
    Returns (cases tested, mismatches)."""
    rng = np.random.default_rng(seed)
    bad = tested = 0
    for it in range(iters):
        nreads = int(rng.integers(1, 14)); glen = int(rng.integers(2, 12))
        g = "".join(rng.choice(list("ACGT"), glen))
        reads = []
        for _ in range(nreads):
            if reads and rng.random() < 0.35:
                reads.append(reads[int(rng.integers(0, len(reads)))]); continue
            L = int(rng.integers(0, glen + 1)); s = int(rng.integers(0, glen - L + 1))
            r = list(g[s:s + L])
            for k in range(len(r)):
                if rng.random() < 0.08:
                    r[k] = str(rng.choice(list("ACGTN")))
            reads.append("".join(r))
        for shuffle in (False, True):
            bwt = ebwt(reads, rng, shuffle)
            suf, F = decode(bwt)
            if suf is None:
                continue
            want = lcp_direct(suf)
            got, work = bfs(bwt)
            tested += 1
            if got != want:
                bad += 1
                if bad < 5:
                    print("MISMATCH", reads, shuffle, "".join(bwt), want, got)
            assert work <= len(bwt) + 6            # every enqueued interval owns an LCP entry
    return tested, bad


if __name__ == "__main__":
    print("tested %d, mismatches %d" % check(int(sys.argv[1]) if len(sys.argv) > 1 else 1, 3000))
