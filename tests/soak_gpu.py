"""Randomised GPU-vs-oracle soak: many small / medium collections of very different shapes (random sets with N and
duplicates, the synthetic generator, homopolymers and tandem repeats, a few very long reads, tiny genomes at huge
coverage, empty and one-base reads) under random parameters (-k -m -v -f -t, M 0-3, B 0-1).  Not collected by pytest:
    python tests/soak_gpu.py <seconds> <first seed>
tests/test_gpu_parity.py::test_randomised_shapes_and_parameters runs the first cases of it.  Round 1: 33 363 cases /
8.96 G rows in nine runs (one of them with BFQ_HUGE_CAP=20000, i.e. batching and the one-workgroup fallback of the
huge-segment rounds), all bit-exact (eBWT, permuted QS, LCP, output reads, statistics, bfq_int mode).
Round 2 adds per case: step 1 in one piece or pile by pile (drawn at random; BFQ_PILES_SPLIT=1 in the environment splits
every pile again), bfq_int mode = LCP deduced from the BWT alone (k_bfs.hip), on small cases also with the ties of
identical suffixes shuffled, the FASTQ job (text in, FASTQ text + streams out) against the oracle's reads, and the global mode
(parallel.run_global on one rank: two-symbol piles, position-mode clusters) against the same.
Since the stream codec exists every case also checks step 5 (containers = CPU statement, eBWT-domain containers back to the streams).
Round 2 totals: 31 992 cases / 8.4 G rows in fifteen runs (the last ten with the 40-bit sort key, the last three with the
step-5 checks, one of those pile by pile), all bit-exact.
Round 3 adds per case: the capped mode (bfq_params.piles = 2: run_reads and the FASTQ job), the global mode's streams
(written through the mapped output files), steps 2-4 without the LF table (k_compact.hip: both jobs, random ring / window
sizes) and, through the step-5 check, the read-order DNA container (BFQDNAC1) on every stream of 64 KiB and more.
Round 3 totals: 7 866 cases / 1.9 G rows in nine runs (the last six -- 4 779 cases -- with the compact mode and the new
containers, two of them with BFQ_PILES_SPLIT=1, one with BFQ_HUGE_CAP=20000), all bit-exact."""
import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bfqzip_amd import api
from tests import util

ACGT = np.array(list(b"ACGT"), np.uint8)


def gen(rng):
    kind = rng.integers(0, 6)
    if kind == 0:      # random small sets with N and duplicates
        n = int(rng.integers(1, 400)); lmax = int(rng.integers(1, 80))
        return util.random_reads(rng, n, 0, lmax, p_n=float(rng.random() * 0.2), dup=float(rng.random() * 0.5))
    if kind == 1:      # synthetic generator, fixed or variable length
        N = int(rng.integers(100, 20000)); L = int(rng.integers(20, 160))
        kw = dict(seed=int(rng.integers(1, 1 << 30)), coverage=int(rng.integers(5, 60)), err_ppm=int(rng.integers(0, 40000)),
                  n_ppm=int(rng.integers(0, 20000)), snp_every=int(rng.integers(50, 2000)), dsnp_every=int(rng.integers(100, 20000)))
        sp = api.synth_spec(N, L, Lmax=int(L + rng.integers(0, 60)) if rng.random() < 0.5 else None, **kw)
        return api.synth_host(sp)
    reads = []
    if kind == 2:      # low complexity: homopolymers / short tandem repeats with noise
        for _ in range(int(rng.integers(50, 6000))):
            L = int(rng.integers(1, 120)); unit = ACGT[rng.integers(0, 4, int(rng.integers(1, 4)))]
            s = np.resize(unit, L).copy()
            e = rng.random(L) < rng.random() * 0.03; s[e] = ACGT[rng.integers(0, 4, int(e.sum()))]
            s[rng.random(L) < 0.003] = ord("N")
            reads.append(s)
    elif kind == 3:    # a few very long reads sharing long stretches
        g = ACGT[rng.integers(0, 4, 30000)]
        for _ in range(int(rng.integers(1, 12))):
            a = int(rng.integers(0, 20000)); reads.append(g[a:a + int(rng.integers(1, 10000))].copy())
    elif kind == 4:    # tiny genome, huge coverage: long clusters, two-symbol sites
        g = ACGT[rng.integers(0, 4, int(rng.integers(30, 300)))]
        g2 = g.copy(); p = rng.integers(0, len(g), max(1, len(g) // 40)); g2[p] = ACGT[rng.integers(0, 4, len(p))]
        for _ in range(int(rng.integers(200, 8000))):
            src = g if rng.random() < 0.5 else g2
            L = int(rng.integers(1, len(g) + 1)); a = int(rng.integers(0, len(g) - L + 1))
            s = src[a:a + L].copy()
            e = rng.random(L) < 0.01; s[e] = ACGT[rng.integers(0, 4, int(e.sum()))]
            reads.append(s)
    else:              # many empty / one-base reads mixed with ordinary ones
        g = ACGT[rng.integers(0, 4, 500)]
        for _ in range(int(rng.integers(1, 3000))):
            r = rng.random()
            if r < 0.3: reads.append(np.zeros(0, np.uint8))
            elif r < 0.5: reads.append(ACGT[rng.integers(0, 4, 1)])
            else:
                a = int(rng.integers(0, 400)); reads.append(g[a:a + int(rng.integers(2, 100))].copy())
    b = np.concatenate(reads) if sum(len(x) for x in reads) else np.zeros(0, np.uint8)
    qlo = int(rng.integers(33, 60)); q = rng.integers(qlo, int(rng.integers(qlo + 1, 127)), len(b)).astype(np.uint8)
    r = np.zeros(len(reads) + 1, np.uint64); r[1:] = np.cumsum([len(x) for x in reads])
    return b, q, r


def run_case(eng, O, seed):
    """One random case; returns (ok, rows, description)."""
    rng = np.random.default_rng(seed)
    b, q, r = gen(rng)
    par = dict(k=int(rng.integers(1, 40)), m=int(rng.integers(1, 9)), v=int(rng.integers(33, 100)), f=int(rng.integers(34, 101)),
               t=int(rng.integers(0, 45)), M=int(rng.integers(0, 4)), B=int(rng.integers(0, 2)))
    piles = int(rng.integers(0, 2))
    eng.set_params(piles=piles, **par)
    p = O.params(K=par["k"], m=par["m"], v=par["v"], f=par["f"], t=par["t"], M=par["M"], B=par["B"])
    bwt, qs, lcp = O.build_ebwt(b, q, r)
    gb, gq, gl = eng.build_ebwt(b, q, r)
    ok = np.array_equal(gb, bwt) and np.array_equal(gq, qs) and np.array_equal(gl.astype(np.uint32), lcp)
    ob, oq, st = O.run_reads(b, q, r, p)
    hb, hq, hst = eng.run_reads(b, q, r)
    ok = ok and np.array_equal(hb, ob) and np.array_equal(hq, oq) and all(st[k] == hst[k] for k in st)
    if par["k"] >= 1:                                         # round 3: the capped mode (workspace cap: two-symbol piles, position-mode
        eng.set_params(piles=2, **par)                        # clusters, no eBWT, no inversion) = the same reads and statistics
        cb, cq, cst = eng.run_reads(b, q, r)
        ok = ok and np.array_equal(cb, ob) and np.array_equal(cq, oq) and all(st[k] == cst[k] for k in st)
        if len(bwt) <= 200000:
            from bfqzip_amd import fastq as _fq
            cres = eng.fastq_job([_fq.format_fastq(b, q, r)], fastq=True, streams=True)
            ok = ok and cres.fastq.tobytes() == _fq.format_fastq(ob, oq, r) and cres.dna.tobytes() == _fq.format_lines(ob, r) \
                and cres.qs.tobytes() == _fq.format_lines(oq, r)
        eng.set_params(piles=piles, **par)
    if len(bwt):
        sb, sq, sroff, sst = eng.smooth_invert(bwt, qs)
        ok = ok and np.array_equal(sb, ob) and np.array_equal(sq, oq) and np.array_equal(sroff, r)
        # round 3: the same two jobs without the LF table (k_compact.hip: what runs under a workspace cap) -- bfq_int's with
        # a ring queue of a random size (levels in chunks / the queue moved to host memory), bfq_ext's with a random LCP window
        os.environ["BFQ_COMPACT"] = "1"
        os.environ["BFQ_COMPACT_RING"] = str(int(rng.choice([64, 300, 5000, 1 << 20])))
        os.environ["BFQ_COMPACT_WIN"] = str(int(rng.choice([100, 5000, 1 << 22])))
        try:
            eng.set_params(piles=piles, **par)                    # (the environment is read when parameters are set)
            cb, cq, croff, cst = eng.smooth_invert(bwt, qs)
            ok = ok and np.array_equal(cb, ob) and np.array_equal(cq, oq) and np.array_equal(croff, r) and all(sst[k] == cst[k] for k in sst)
            if int(lcp.max(initial=0)) < 65536:
                xb, xq, xroff, xst = eng.smooth_invert(bwt, qs, lcp.astype(np.uint16 if rng.integers(0, 2) else np.uint32))
                ok = ok and np.array_equal(xb, ob) and np.array_equal(xq, oq) and np.array_equal(xroff, r) and all(sst[k] == xst[k] for k in sst)
        finally:
            for k in ("BFQ_COMPACT", "BFQ_COMPACT_RING", "BFQ_COMPACT_WIN"):
                del os.environ[k]
            eng.set_params(piles=piles, **par)
    if 0 < len(bwt) <= 4000:                                  # any tie order: the reference sees one terminator symbol
        tb, tq = util.shuffle_ties(bwt, qs, rng)
        eb, eq, eroff, est = O.smooth_invert(tb, tq, None, p)
        sb, sq, sroff, sst = eng.smooth_invert(tb, tq)
        ok = ok and np.array_equal(sb, eb) and np.array_equal(sq, eq) and np.array_equal(sroff, eroff) and all(est[k] == sst[k] for k in est)
    if len(bwt) <= 200000:                                    # FASTQ text in -> FASTQ text + line streams out
        from bfqzip_amd import fastq
        text = fastq.format_fastq(b, q, r)
        res = eng.fastq_job([text], fastq=True, streams=True)
        ok = ok and res.fastq.tobytes() == fastq.format_fastq(ob, oq, r) and res.dna.tobytes() == fastq.format_lines(ob, r) \
            and res.qs.tobytes() == fastq.format_lines(oq, r)
        # step 5: the streams through the codec (= the CPU statement, byte for byte), and the eBWT-domain containers
        # (rows of the edited eBWT; qualities by row or by read) back to the same streams
        for mode in (1, 2, 3):
            z = eng.fastq_job([text], fastq=False, streams=True, compress=mode)
            if mode == 1:
                ok = ok and np.array_equal(np.asarray(z.dna), O.codec_encode(np.asarray(res.dna))) \
                    and np.array_equal(np.asarray(eng.stream_decompress(np.asarray(z.qs))), np.asarray(res.qs))
            else:
                d2, q2, nr = eng.ebwt_decode(np.asarray(z.dna), np.asarray(z.qs))
                ok = ok and nr == len(r) - 1 and np.array_equal(d2, np.asarray(res.dna)) and np.array_equal(q2, np.asarray(res.qs))
        if par["k"] >= 2 and len(r) > 1:                         # global mode (one eBWT dealt pile by pile, position-mode clusters) = the same reads
            import tempfile
            from bfqzip_amd import parallel
            with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
                open(d + "/in.fastq", "wb").write(text)
                names = parallel.output_names([d + "/in.fastq"], d + "/G", False)
                tot = parallel.run_global(eng, parallel.Comm(), [d + "/in.fastq"], names, want_streams=True)
                ok = ok and open(names[0]["fastq"], "rb").read() == res.fastq.tobytes() \
                    and open(names[0]["dna"], "rb").read() == res.dna.tobytes() and open(names[0]["qs"], "rb").read() == res.qs.tobytes() \
                    and all(tot["stats_all_ranks"][k] == st[k] for k in ("num_clust", "qs_smoothed", "modified", "num_clust_mod", "bases_inside"))
    return ok, len(bwt), "seed %d %s piles %d reads %d rows %d" % (seed, par, piles, len(r) - 1, len(bwt))


def main(seconds, seed0):
    from oracle import orc as O
    eng = api.Engine(0)
    t0 = time.time(); it = 0; rows = 0
    while time.time() - t0 < seconds:
        ok, n, what = run_case(eng, O, seed0 + it)
        rows += n
        if not ok:
            print("MISMATCH", what, flush=True)
            return 1
        it += 1
        if it % 50 == 0:
            print("cases", it, "rows", rows, "elapsed %.0f s" % (time.time() - t0), flush=True)
    print("SOAK OK cases", it, "rows", rows)
    return 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]), int(sys.argv[2])))
