"""GPU parity tests: libbfqhip.so (through its C-ABI) against the CPU oracle and the
golden vectors of the compiled reference.  Bit-exact: integer / byte / index work;
the only floating point (M=1) is evaluated through host-libm tables and must be
bit-exact as well."""
import os
import numpy as np
import pytest
from bfqzip_amd import api, fastq
from tests import util

pytestmark = pytest.mark.gpu
IDX = util.golden_index()
CASES = [(name, key) for name in IDX for key in IDX[name]["out"]]


def _engine_params(d):
    return dict(k=d["k"], m=d["m"], v=d["v"], f=d["f"], t=d["t"], M=d["M"], B=d["B"])


@pytest.mark.parametrize("name", list(IDX))
def test_build_ebwt_matches_golden(engine, name):
    b, q, r, h, bwt, qs, lcp = util.golden_set(name)
    bwt2, qs2, lcp2 = engine.build_ebwt(b, q, r)
    assert np.array_equal(bwt2, bwt)
    assert np.array_equal(qs2, qs)
    assert np.array_equal(lcp2, lcp)


@pytest.mark.parametrize("name,key", CASES)
def test_smooth_invert_matches_reference_output(engine, name, key):
    b, q, r, h, bwt, qs, lcp = util.golden_set(name)
    d, hdr = util.parse_case(key)
    engine.set_params(**_engine_params(d))
    ob, oq, oroff, st = engine.smooth_invert(bwt, qs)                 # bfq_int mode: LCP deduced
    assert util.md5(fastq.format_fastq(ob, oq, oroff, h if hdr else None)) == IDX[name]["out"][key]
    ob2, oq2, oroff2, st2 = engine.smooth_invert(bwt, qs, lcp)       # bfq_ext mode: LCP given
    assert np.array_equal(ob, ob2) and np.array_equal(oq, oq2) and np.array_equal(oroff, oroff2)
    ob3, oq3, st3 = engine.run_reads(b, q, r)                        # fused
    assert np.array_equal(ob, ob3) and np.array_equal(oq, oq3)
    for k in ("num_clust", "bases_inside", "qs_smoothed", "modified"):
        assert st[k] == st2[k] == st3[k]


def _check_against_oracle(engine, orc, b, q, r, **par):
    full = dict(k=16, m=2, v=ord(">"), f=40, t=20, M=2, B=0)
    full.update(par)
    engine.set_params(**full)
    full.pop("piles", None)
    p = orc.params(K=full["k"], m=full["m"], v=full["v"], f=full["f"], t=full["t"], M=full["M"], B=full["B"])
    bwt, qs, lcp = orc.build_ebwt(b, q, r)
    gb, gq, gl = engine.build_ebwt(b, q, r)
    assert np.array_equal(gb, bwt), "bwt"
    assert np.array_equal(gq, qs), "qs"
    assert np.array_equal(gl.astype(np.uint32), lcp), "lcp"
    ob, oq, st = orc.run_reads(b, q, r, p)
    hb, hq, hst = engine.run_reads(b, q, r)
    assert np.array_equal(hb, ob), "bases"
    assert np.array_equal(hq, oq), "quals"
    for k in st:
        assert st[k] == hst[k], k
    if len(bwt):
        sb, sq, sroff, sst = engine.smooth_invert(bwt, qs)
        assert np.array_equal(sb, ob) and np.array_equal(sq, oq) and np.array_equal(sroff, r)
    return hst


@pytest.mark.parametrize("M,B", [(M, B) for M in range(4) for B in range(2)])
def test_synthetic_all_modes(engine, orc, M, B):
    sp = api.synth_spec(3000, 50, seed=100 + M * 2 + B, coverage=25)
    b, q, r = api.synth_host(sp)
    st = _check_against_oracle(engine, orc, b, q, r, M=M, B=B, m=5)
    assert st["num_clust"] > 100


def test_variable_length_two_symbol_branch(engine, orc):
    sp = api.synth_spec(4000, 20, Lmax=70, seed=5, coverage=40, err_ppm=20000, n_ppm=15000, snp_every=97, dsnp_every=131)
    b, q, r = api.synth_host(sp)
    st = _check_against_oracle(engine, orc, b, q, r, m=5)
    assert st["num_clust_mod"] > 0            # the two-frequent-symbol "processed" branch ran


def test_random_small_sets(engine, orc):
    rng = np.random.default_rng(2024)
    for it in range(40):
        nreads = int(rng.integers(1, 300)); lmax = int(rng.integers(1, 60))
        b, q, r = util.random_reads(rng, nreads, 1, lmax)
        _check_against_oracle(engine, orc, b, q, r, M=int(rng.integers(0, 4)), B=int(rng.integers(0, 2)),
                              k=int(rng.choice([1, 2, 3, 5, 8, 16])), m=int(rng.choice([2, 3, 5, 9])),
                              v=int(rng.choice([62, 53, 73])), t=int(rng.choice([5, 20, 35])),
                              f=int(rng.choice([40, 50, 70])))


def test_edge_cases(engine, orc):
    A = lambda s: np.frombuffer(s, np.uint8)
    one = (A(b"ACGTN"), A(b"IIII#"), np.array([0, 5], np.uint64))
    single_base = (A(b"A"), A(b"I"), np.array([0, 1], np.uint64))
    with_empty = (A(b"ACGTAC"), A(b"IIIIII"), np.array([0, 3, 3, 6], np.uint64))     # an empty read in the middle
    # lengths 1 and 3: n - N is a multiple of N although the reads differ in length (the equal-length guess of
    # bfq_smooth_invert must notice and fall back to counting)
    uneven = (A(b"AACG"), A(b"IIII"), np.array([0, 1, 4], np.uint64))
    for b, q, r in (one, single_base, with_empty, uneven):
        _check_against_oracle(engine, orc, b, q, r, m=2, k=1)
    # many identical reads: segments far larger than a wavefront (k_refine_big), ties by read index
    rd = A(b"ACGTTGCAACGTACGTTTGACCAGTACGATCGATCGTAGCTAGCTAGCATCGATCAGCTACGATCGATCAGCATCGA")
    nrep = 300
    b = np.tile(rd, nrep); q = np.tile(np.arange(len(rd), dtype=np.uint8) % 40 + 35, nrep)
    r = np.arange(nrep + 1, dtype=np.uint64) * len(rd)
    st = _check_against_oracle(engine, orc, b, q, r, m=5)
    assert st["n_big_segments"] > 0
    # low complexity: poly-A of different lengths plus a few N-only reads
    reads = [b"A" * L for L in (1, 2, 30, 64, 65, 100, 100, 100, 7)] + [b"N" * 5, b"NNNNNNNNNNNNNNNNNNNNNNNNN"] + [b"A" * 90] * 70
    b = A(b"".join(reads)); q = np.full(len(b), 70, np.uint8)
    r = np.zeros(len(reads) + 1, np.uint64); r[1:] = np.cumsum([len(x) for x in reads])
    _check_against_oracle(engine, orc, b, q, r, m=2, k=3)


def test_errors_are_loud(engine):
    A = lambda s: np.frombuffer(s, np.uint8)
    with pytest.raises(api.BfqError) as e:
        engine.run_reads(A(b"ACGX"), A(b"IIII"), np.array([0, 4], np.uint64))
    assert e.value.code == -3
    with pytest.raises(api.BfqError):
        engine.smooth_invert(A(b"AC#G"), A(b"IIII"))          # not an eBWT in suffix order


def test_device_resident_and_synth_device(engine, orc):
    torch = pytest.importorskip("torch")
    sp = api.synth_spec(20000, 100, seed=77)
    hb, hq, hr = api.synth_host(sp)
    total = len(hb)
    dev = torch.device("cuda:0")
    db = torch.empty(total, dtype=torch.uint8, device=dev); dq = torch.empty_like(db)
    dr = torch.empty(sp.N + 1, dtype=torch.int64, device=dev)
    engine.synth_device(sp, db.data_ptr(), dq.data_ptr(), dr.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(db.cpu().numpy(), hb) and np.array_equal(dq.cpu().numpy(), hq)
    assert np.array_equal(dr.cpu().numpy().astype(np.uint64), hr)
    ob = torch.empty_like(db); oq = torch.empty_like(db)
    engine.set_params(m=5)
    st = engine.run_reads_device(db.data_ptr(), dq.data_ptr(), dr.data_ptr(), sp.N, total, ob.data_ptr(), oq.data_ptr())
    torch.cuda.synchronize()
    eb, eq, est = orc.run_reads(hb, hq, hr, orc.params(m=5))
    assert np.array_equal(ob.cpu().numpy(), eb) and np.array_equal(oq.cpu().numpy(), eq)
    for k in est:
        assert est[k] == st[k]
    prof = engine.prof()
    assert "k_radix_scatter" in prof and prof["k_radix_scatter"]["launches"] >= 8


def test_bfq_ext_arithmetic(engine, orc):
    """bfq_ext's M=3 rounding (round((float)sum/num), bfq_ext.cpp:496) and uchar M=1 result, as the
    oracle restates them (the reference bfq_ext cannot be built here without stand-ins: unpinned)."""
    sp = api.synth_spec(3000, 50, seed=321, coverage=25)
    b, q, r = api.synth_host(sp)
    bwt, qs, lcp = orc.build_ebwt(b, q, r)
    for M in (1, 3):
        engine.set_params(m=5, M=M, ext=1)
        p = orc.params(m=5, M=M, ext=1)
        ob, oq, oroff, st = orc.smooth_invert(bwt, qs, lcp, p)
        gb, gq, groff, gst = engine.smooth_invert(bwt, qs, np.minimum(lcp, 255).astype(np.uint8))   # 1-byte LCP file (--lbytes 1)
        assert np.array_equal(gb, ob) and np.array_equal(gq, oq) and np.array_equal(groff, oroff)
        assert gst["qs_smoothed"] == st["qs_smoothed"]
    # the two roundings really differ somewhere
    engine.set_params(m=5, M=3, ext=0)
    ib, iq, _, _ = engine.smooth_invert(bwt, qs, lcp.astype(np.uint16))
    assert not np.array_equal(iq, gq)


def test_pile_mode(engine, orc, monkeypatch):
    """Step 1 one first-symbol pile at a time (k_piles.hip; bfq_params.piles = 1): same eBWT / QS / LCP, same reads, same
    statistics as the oracle -- synthetic sets in all smoothing modes, random small sets (variable length, duplicates, N,
    empty reads), low-complexity reads (huge segments inside one pile), and the same with every pile split again by its
    second symbol (BFQ_PILES_SPLIT: what a pile beyond the workspace goes through)."""
    def cases():
        for M in range(4):
            sp = api.synth_spec(3000, 50, seed=300 + M, coverage=25)
            yield api.synth_host(sp) + (dict(M=M, B=M & 1, m=5),)
        rng = np.random.default_rng(77)
        for it in range(12):
            b, q, r = util.random_reads(rng, int(rng.integers(1, 300)), 0 if it % 4 == 0 else 1, int(rng.integers(1, 60)))
            yield b, q, r, dict(M=int(rng.integers(0, 4)), k=int(rng.choice([1, 2, 3, 5, 16])), m=int(rng.choice([2, 3, 5])))
        yield _low_complexity(np.random.default_rng(7)) + (dict(m=5),)
        A = lambda s: np.frombuffer(s, np.uint8)
        yield A(b"ACGTN"), A(b"IIII#"), np.array([0, 5], np.uint64), dict(m=2, k=1)
        yield A(b""), A(b""), np.array([0, 0, 0], np.uint64), dict(m=2, k=1)              # only empty reads
    for split in (False, True):
        if split:
            monkeypatch.setenv("BFQ_PILES_SPLIT", "1")
        for b, q, r, par in cases():
            st = _check_against_oracle(engine, orc, b, q, r, piles=1, **par)
    monkeypatch.delenv("BFQ_PILES_SPLIT")
    assert st is not None
    engine.set_params()


def test_capped_mode(engine, orc):
    """The whole path under a workspace cap (bfq_params.ws_cap_mib / piles = 2): two-symbol piles one at a
    time, position-mode clusters, no eBWT, no LF table, no inversion -- the same reads and statistics as the oracle in
    every smoothing mode, on variable-length / duplicate / N / empty reads, on low-complexity reads (huge segments and
    clusters inside one pile), through the host-array call and through the FASTQ job (text, streams, headers); -k 1 runs
    on first-symbol piles, -k 0 is refused; a 2 M-read block under a cap that rules out both other modes equals its
    uncapped run."""
    from bfqzip_amd import fastq as fq

    def cases():
        for M in range(4):
            sp = api.synth_spec(3000, 50, seed=400 + M, coverage=25)
            yield api.synth_host(sp) + (dict(M=M, B=M & 1, m=5),)
        yield api.synth_host(api.synth_spec(4000, 20, Lmax=70, seed=5, coverage=40, err_ppm=20000, n_ppm=15000, snp_every=97, dsnp_every=131)) + (dict(m=5),)
        rng = np.random.default_rng(78)
        for it in range(12):
            b, q, r = util.random_reads(rng, int(rng.integers(1, 300)), 0 if it % 4 == 0 else 1, int(rng.integers(1, 60)))
            yield b, q, r, dict(M=int(rng.integers(0, 4)), B=int(rng.integers(0, 2)), k=int(rng.choice([1, 2, 3, 5, 16])), m=int(rng.choice([2, 3, 5])))
        yield _low_complexity(np.random.default_rng(7)) + (dict(m=5, M=1, B=1),)
        A = lambda s: np.frombuffer(s, np.uint8)
        yield A(b"ACGTN"), A(b"IIII#"), np.array([0, 5], np.uint64), dict(m=2, k=2)
        yield A(b""), A(b""), np.array([0, 0, 0], np.uint64), dict(m=2, k=2)              # only empty reads
    two = 0
    for ci, (b, q, r, par) in enumerate(cases()):
        full = dict(k=16, m=2, v=ord(">"), f=40, t=20, M=2, B=0); full.update(par)
        print(f"capped case {ci}: {len(r) - 1} reads, {len(b)} bases, {par}", flush=True)
        p = orc.params(K=full["k"], m=full["m"], v=full["v"], f=full["f"], t=full["t"], M=full["M"], B=full["B"])
        ob, oq, ost = orc.run_reads(b, q, r, p)
        engine.set_params(piles=2, **full)
        gb, gq, gst = engine.run_reads(b, q, r)
        assert np.array_equal(gb, ob) and np.array_equal(gq, oq), par
        for k in ost:
            assert ost[k] == gst[k], (k, par)
        two += ost["num_clust_mod"]
        hdrs = [b"@r%d some text" % i for i in range(len(r) - 1)]
        text = fq.format_fastq(b, q, r, hdrs)
        res = engine.fastq_job([text], keep_headers=True, fastq=True, streams=True, hdr=True)
        assert np.asarray(res.fastq).tobytes() == fq.format_fastq(ob, oq, r, hdrs)
        assert np.asarray(res.dna).tobytes() == fq.format_lines(ob, r) and np.asarray(res.qs).tobytes() == fq.format_lines(oq, r)
        assert np.asarray(res.hdr).tobytes() == b"".join(h + b"\n" for h in hdrs)
        only = engine.fastq_job([text], keep_headers=False, fastq=True, streams=False)
        assert np.asarray(only.fastq).tobytes() == fq.format_fastq(ob, oq, r)
    assert two > 0
    b, q, r = api.synth_host(api.synth_spec(2000, 40, seed=9))
    engine.set_params(piles=2, k=0)
    with pytest.raises(api.BfqError):
        engine.run_reads(b, q, r)
    # a block whose one-piece (7 GB) and pile-by-pile (3.9 GB) workspaces are above the cap: the cap selects the mode
    sp = api.synth_spec(2_000_000, 100, seed=12)
    b, q, r = api.synth_host(sp)
    engine.set_params(m=5, M=1, B=1)
    fb, fq_, fst = engine.run_reads(b, q, r)
    engine.set_params(m=5, M=1, B=1, ws_cap_mib=3000)
    cb, cq, cst = engine.run_reads(b, q, r)
    assert engine.workspace_bytes() <= 3000 << 20
    assert np.array_equal(cb, fb) and np.array_equal(cq, fq_)
    assert {k: cst[k] for k in cst if k != "n_big_segments"} == {k: fst[k] for k in fst if k != "n_big_segments"}
    with pytest.raises(api.BfqError) as e:                                # the eBWT arrays themselves do not fit under it
        engine.build_ebwt(b, q, r)
    assert "cap" in str(e.value)
    engine.set_params(m=5, ws_cap_mib=300)                               # and a cap nothing fits under is an error, not a crash
    with pytest.raises(api.BfqError):
        engine.run_reads(b, q, r)
    engine.set_params()


def test_compact_steps_on_a_given_ebwt(engine, orc, monkeypatch):
    """bfq_ext's job (eBWT + QS + LCP given) without the LF table (k_compact.hip: rank blocks answered on demand, qualities
    smoothed in place, replacement array, the LCP streamed through a window): what runs when the workspace cap rules the
    17 bytes per row of the table out.  Same reads, offsets and statistics as the oracle for every smoothing mode (incl. the
    bfq_ext arithmetic), 1- / 2- / 4-byte LCP entries, variable-length / empty / N reads, low-complexity reads (clusters of
    10^5 rows through the tiled kernels), a window of 1000 rows (the LCP in many pieces), and through the FASTQ-writing entry
    point with a header file; bfq_int's job too (no LCP: deduced on the same rank blocks with a ring queue, dropped once the
    flags exist).  Then a 2 M-read eBWT under caps that leave no room for the table, in both modes."""
    monkeypatch.setenv("BFQ_COMPACT", "1")

    def cases():
        for M in range(4):
            for ext in (0, 1):
                sp = api.synth_spec(3000, 50, seed=500 + 2 * M + ext, coverage=25)
                yield api.synth_host(sp) + (dict(M=M, B=(M + ext) & 1, m=5, ext=ext),)
        yield api.synth_host(api.synth_spec(4000, 20, Lmax=70, seed=5, coverage=40, err_ppm=20000, n_ppm=15000, snp_every=97, dsnp_every=131)) + (dict(m=5),)
        rng = np.random.default_rng(79)
        for it in range(12):
            b, q, r = util.random_reads(rng, int(rng.integers(1, 300)), 0 if it % 4 == 0 else 1, int(rng.integers(1, 60)))
            yield b, q, r, dict(M=int(rng.integers(0, 4)), B=int(rng.integers(0, 2)), k=int(rng.choice([0, 1, 2, 3, 5, 16])), m=int(rng.choice([2, 3, 5])))
        yield _low_complexity(np.random.default_rng(7)) + (dict(m=5, M=1, B=1),)
    two = 0
    for ci, (b, q, r, par) in enumerate(cases()):
        full = dict(k=16, m=2, v=ord(">"), f=40, t=20, M=2, B=0, ext=0); full.update(par)
        p = orc.params(K=full["k"], m=full["m"], v=full["v"], f=full["f"], t=full["t"], M=full["M"], B=full["B"], ext=full["ext"])
        bwt, qs, lcp = orc.build_ebwt(b, q, r)
        if not len(bwt):
            continue
        eb, eq, eroff, est = orc.smooth_invert(bwt, qs, lcp.astype(np.uint32), p)
        if ci % 3 == 1:
            monkeypatch.setenv("BFQ_COMPACT_WIN", "1000")
        else:
            monkeypatch.delenv("BFQ_COMPACT_WIN", raising=False)
        engine.set_params(**full)
        for dt in (np.uint8, np.uint16, np.uint32)[ci % 3:][:2]:
            if dt == np.uint8 and int(lcp.max(initial=0)) > 255:
                continue
            gb, gq, groff, gst = engine.smooth_invert(bwt, qs, lcp.astype(dt))
            assert np.array_equal(gb, eb) and np.array_equal(gq, eq) and np.array_equal(groff, eroff), (ci, par, dt)
            for k in est:
                assert est[k] == gst[k], (k, ci, par)
        # bfq_int mode; every other case with a ring of 64 entries: levels in chunks, then the queue in host memory
        if ci % 2:
            monkeypatch.setenv("BFQ_COMPACT_RING", "64")
            engine.set_params(**full)                                 # (the environment is read when parameters are set)
        ib, iq, iroff, ist = engine.smooth_invert(bwt, qs)
        monkeypatch.delenv("BFQ_COMPACT_RING", raising=False)
        assert np.array_equal(ib, eb) and np.array_equal(iq, eq) and np.array_equal(iroff, eroff), (ci, par)
        for k in est:
            assert est[k] == ist[k], (k, ci, par)
        two += est["num_clust_mod"]
        hdrs = b"".join(b"@h%d\n" % i for i in range(len(r) - 1))
        txt, _ = engine.smooth_invert_fastq(bwt, qs, lcp=lcp.astype(np.uint32), headers=np.frombuffer(hdrs, np.uint8))
        from bfqzip_amd import fastq as fq
        assert txt == fq.format_fastq(eb, eq, eroff, [b"@h%d" % i for i in range(len(r) - 1)])
    assert two > 0
    monkeypatch.delenv("BFQ_COMPACT"); monkeypatch.delenv("BFQ_COMPACT_WIN", raising=False)
    # the cap selects it: 2 M x 100 (n = 202 M rows: the table path needs 3.7 GB, the compact one 1.6 GB)
    sp = api.synth_spec(2_000_000, 100, seed=12)
    b, q, r = api.synth_host(sp)
    engine.set_params(m=5, M=1, B=1)
    bwt, qs, lcp = engine.build_ebwt(b, q, r)
    fb, fq_, froff, fst = engine.smooth_invert(bwt, qs, lcp)
    engine.set_params(m=5, M=1, B=1, ws_cap_mib=2000)
    cb, cq, croff, cst = engine.smooth_invert(bwt, qs, lcp)
    assert engine.workspace_bytes() <= 2000 << 20
    assert np.array_equal(cb, fb) and np.array_equal(cq, fq_) and np.array_equal(croff, froff) and cst == fst
    engine.set_params(m=5, M=1, B=1, ws_cap_mib=2600)                 # bfq_int mode: + the 16-bit LCP and the ring while the flags are made
    ib, iq, iroff, ist = engine.smooth_invert(bwt, qs)
    assert engine.workspace_bytes() <= 2600 << 20
    assert np.array_equal(ib, fb) and np.array_equal(iq, fq_) and np.array_equal(iroff, froff) and ist == fst
    # rings smaller than a level and its children (the default takes what the cap leaves): levels launched in chunks as
    # their parents' slots come free (n / 5), the queue moved to host memory when two levels do not fit at all (n / 40)
    for div in (5, 40):
        monkeypatch.setenv("BFQ_COMPACT_RING", str(len(bwt) // div))
        engine.set_params(m=5, M=1, B=1, ws_cap_mib=2600)
        ib, iq, iroff, ist = engine.smooth_invert(bwt, qs)
        assert np.array_equal(ib, fb) and np.array_equal(iq, fq_) and np.array_equal(iroff, froff) and ist == fst, div
    monkeypatch.delenv("BFQ_COMPACT_RING")
    engine.set_params(m=5, ws_cap_mib=400)
    with pytest.raises(api.BfqError) as e:
        engine.smooth_invert(bwt, qs, lcp)
    assert "cap" in str(e.value)
    engine.set_params()


def test_position_mode(engine, orc, monkeypatch):
    """The device-resident fused path without LF table and walks (BFQ_POSMODE=1: k_cluster writes its edits to the text
    position every row's sort record carries): same reads and statistics as the oracle -- all smoothing modes with and without
    binning, the two-frequent-symbol branch (the symbol before the preceding one comes from the packed text), clusters of
    10^5 rows (k_big_*), variable-length and empty reads."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("BFQ_POSMODE", "1")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(31337)
    sets = []
    for M in range(4):
        for B in range(2):
            sets.append(api.synth_host(api.synth_spec(3000, 50, seed=700 + 2 * M + B, coverage=25)) + (dict(M=M, B=B, m=5),))
    sets.append(api.synth_host(api.synth_spec(4000, 20, Lmax=70, seed=5, coverage=40, err_ppm=20000, n_ppm=15000, snp_every=97, dsnp_every=131)) + (dict(m=5),))
    sets.append(_low_complexity(np.random.default_rng(7)) + (dict(m=5, M=1, B=1),))
    for it in range(10):
        b, q, r = util.random_reads(rng, int(rng.integers(1, 300)), 0 if it % 3 == 0 else 1, int(rng.integers(1, 60)))
        sets.append((b, q, r, dict(M=int(rng.integers(0, 4)), B=int(rng.integers(0, 2)), k=int(rng.choice([1, 2, 5, 16])), m=int(rng.choice([2, 5])))))
    two = 0
    for b, q, r, par in sets:
        full = dict(k=16, m=2, v=ord(">"), f=40, t=20, M=2, B=0); full.update(par)
        engine.set_params(**full)
        p = orc.params(K=full["k"], m=full["m"], v=full["v"], f=full["f"], t=full["t"], M=full["M"], B=full["B"])
        eb, eq, est = orc.run_reads(b, q, r, p)
        n = max(len(b), 1)
        db = torch.zeros(n, dtype=torch.uint8, device=dev); dq = torch.zeros_like(db)
        db[:len(b)] = torch.from_numpy(b.copy()).to(dev); dq[:len(b)] = torch.from_numpy(q.copy()).to(dev)
        dr = torch.from_numpy(r.astype(np.int64)).to(dev)
        ob = torch.zeros_like(db); oq = torch.zeros_like(db)
        st = engine.run_reads_device(db.data_ptr(), dq.data_ptr(), dr.data_ptr(), len(r) - 1, len(b), ob.data_ptr(), oq.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(ob.cpu().numpy()[:len(b)], eb) and np.array_equal(oq.cpu().numpy()[:len(b)], eq), par
        for k in est:
            assert est[k] == st[k], (k, par)
        two += est["num_clust_mod"]
    assert two > 0
    engine.set_params()


def test_bfq_int_mode_any_tie_order(engine, orc):
    """bfq_int mode deduces the LCP from the BWT alone (k_bfs.hip) and, like the reference (one terminator symbol),
    takes identical suffixes -- and the terminator rows -- in ANY order: eBWTs with shuffled ties must give what the
    oracle gives (tests/test_oracle_golden.py pins that behaviour against the compiled reference)."""
    rng = np.random.default_rng(4321)
    done = 0
    for it in range(25):
        b, q, r = util.random_reads(rng, int(rng.integers(5, 120)), 0 if it % 5 == 0 else 1, int(rng.integers(4, 40)), dup=0.5, p_n=0.02)
        bwt, qs, lcp = orc.build_ebwt(b, q, r)
        sb, sq = util.shuffle_ties(bwt, qs, rng)
        done += int(not np.array_equal(sb, bwt))
        M, B = int(rng.integers(0, 4)), int(rng.integers(0, 2))
        k, m = int(rng.choice([1, 2, 3, 5, 8])), int(rng.choice([2, 3, 5]))
        p = orc.params(K=k, m=m, M=M, B=B)
        ob, oq, oroff, st = orc.smooth_invert(sb, sq, None, p)
        engine.set_params(k=k, m=m, M=M, B=B)
        gb, gq, groff, gst = engine.smooth_invert(sb, sq)
        assert np.array_equal(gb, ob) and np.array_equal(gq, oq) and np.array_equal(groff, oroff), it
        for key in st:
            assert st[key] == gst[key], (it, key)
    assert done >= 15
    # blocks of identical suffixes far beyond a wavefront (the fill list), reads that are prefixes of others, empty reads
    rd = np.frombuffer(b"ACGTTGCAACGTACGTTTGACCAGTACGATCG", np.uint8)
    reads = [rd] * 700 + [rd[:9]] * 300 + [rd[3:]] * 150 + [np.zeros(0, np.uint8)] * 40 + [rd[::-1].copy()] * 3
    reads = [reads[i] for i in rng.permutation(len(reads))]
    b = np.concatenate(reads); q = rng.integers(33, 74, len(b)).astype(np.uint8)
    r = np.zeros(len(reads) + 1, np.uint64); r[1:] = np.cumsum([len(x) for x in reads])
    bwt, qs, lcp = orc.build_ebwt(b, q, r)
    for shuffle in (False, True):
        sb, sq = util.shuffle_ties(bwt, qs, rng) if shuffle else (bwt, qs)
        p = orc.params(K=4, m=5)
        ob, oq, oroff, st = orc.smooth_invert(sb, sq, None, p)
        engine.set_params(k=4, m=5)
        gb, gq, groff, gst = engine.smooth_invert(sb, sq)
        assert np.array_equal(gb, ob) and np.array_equal(gq, oq) and np.array_equal(groff, oroff)
    engine.set_params()


def test_ebwt_modes_config2_size(engine, orc):
    """BASELINE configs[1] shape (1 M x 100 bp, n = 101 M rows): the eBWT of the fused run handed back in bfq_int mode (LCP
    deduced from the BWT) and in bfq_ext mode (1-byte LCP file, as eGap --lbytes 1 writes it) against the oracle."""
    sp = api.synth_spec(1_000_000, 100, seed=20240807)
    b, q, r = api.synth_host(sp)
    p = orc.params(m=5)
    eb, eq, est = orc.run_reads(b, q, r, p)
    engine.set_params(m=5)
    gb, gq, gst = engine.run_reads(b, q, r)
    assert np.array_equal(gb, eb) and np.array_equal(gq, eq)
    n = len(b) + len(r) - 1
    bwt, qs, lcp = engine.fetch_ebwt(n)
    ib, iq, iroff, ist = engine.smooth_invert(bwt, qs)                                   # bfq_int mode
    assert np.array_equal(ib, eb) and np.array_equal(iq, eq) and np.array_equal(iroff, r)
    engine.set_params(m=5, ext=1)
    xb, xq, xroff, xst = engine.smooth_invert(bwt, qs, np.minimum(lcp, 255).astype(np.uint8))   # bfq_ext mode
    assert np.array_equal(xb, eb) and np.array_equal(xq, eq) and np.array_equal(xroff, r)
    for k in est:
        assert est[k] == gst[k] == ist[k], k
    engine.set_params()


def test_full_size_properties(engine):
    """BASELINE.json configs[2] size (30 M x 150 bp, n = 4.53 G rows: the only test beyond 2^32 elements) through
    size-independent properties.  Needs 200 GiB of free HBM: skipped -- visibly -- otherwise.
      1. K above every LCP -> no cluster -> the output must be the input (sort + LF round trip);
      2. M=2 B=0: edits consistent with the statistics and with the M=2 rule, deterministic;
      3. B=1 (configs[2] itself): bases as in 2., qualities = Illumina-binned qualities of 2.;
      4. M=1 (configs[4]'s smoothing): same base edits and same smoothed positions as 2. (the decision tree does not
         depend on M), qualities differ only there;
      5. bfq_int mode (LCP deduced from the eBWT alone) on the 4.53 G-row eBWT of run 2: same reads as run 2;
      5b. steps 2-4 on that eBWT under a 40 GiB cap (no LF table: k_compact.hip), with the LCP given and deduced: the same;
      2b. step 1 pile by pile: same reads and statistics as run 2;
      2c. the capped mode under a 40 GiB workspace cap: same reads and statistics as run 2."""
    torch = pytest.importorskip("torch")
    free, total = torch.cuda.mem_get_info()
    if free < 200 * 2**30:
        pytest.skip(f"full-size test needs 200 GiB of free HBM, {free / 2**30:.0f} GiB free")
    N, L = 30_000_000, 150
    print(f"full-size test: {N} x {L}")
    sp = api.synth_spec(N, L, seed=4242)
    tot = N * L
    dev = torch.device("cuda:0")
    db = torch.empty(tot, dtype=torch.uint8, device=dev); dq = torch.empty_like(db)
    dr = torch.empty(N + 1, dtype=torch.int64, device=dev)
    ob = torch.empty_like(db); oq = torch.empty_like(db)
    engine.synth_device(sp, db.data_ptr(), dq.data_ptr(), dr.data_ptr())
    torch.cuda.synchronize()
    assert int(dr[-1].item()) == tot and bool((dr[1:] - dr[:-1] == L).all())
    run = lambda: engine.run_reads_device(db.data_ptr(), dq.data_ptr(), dr.data_ptr(), N, tot, ob.data_ptr(), oq.data_ptr())
    engine.set_params(k=10000, m=5)
    st = run(); torch.cuda.synchronize()
    assert st["num_clust"] == 0 and st["n_rows"] == tot + N and st["n_reads"] == N
    assert torch.equal(ob, db) and torch.equal(oq, dq)                      # 1. identity round trip
    engine.set_params(k=16, m=5, M=2, B=0, v=ord(">"))
    st = run(); torch.cuda.synchronize()
    changed_b = ob != db
    changed_q = oq != dq
    assert int(changed_b.sum().item()) == st["modified"]                   # every replacement changes the base
    assert 0 < int(changed_q.sum().item()) <= st["qs_smoothed"]
    assert bool(((oq == ord(">")) | ~changed_q).all())                     # M=2: smoothed qualities are the constant
    assert not bool((changed_b & changed_q).any())                         # a replaced base keeps its quality
    assert bool(((ob == ord("N")) <= (db == ord("N"))).all())              # N is never written
    st2 = run(); torch.cuda.synchronize()
    assert st2 == st                                                        # 2. deterministic
    # 5. bfq_int mode on this run's eBWT (host arrays: the boundary of bfq_smooth_invert)
    n = tot + N
    bwt, qs, lcp = engine.fetch_ebwt(n)
    ib, iq, iroff, ist = engine.smooth_invert(bwt, qs)
    # 5b. the same two jobs under a 40 GiB workspace cap: no LF table (k_compact.hip) -- bfq_ext's (LCP given, streamed
    #     through a window) and bfq_int's (LCP deduced with a ring queue)
    import time as _t
    for label, l in (("bfq_ext", lcp), ("bfq_int", None)):
        engine.set_params(k=16, m=5, M=2, B=0, v=ord(">"), ws_cap_mib=40 * 1024)
        # timed on the second call: the first one's allocation waits for the driver to scrub the 77 GiB the uncapped call
        # before it has just given back (freed HBM is cleared at ~30 GB/s and the next allocation waits for it)
        dts = []
        for _ in range(2):
            t0 = _t.perf_counter(); cb, cq, croff, cst = engine.smooth_invert(bwt, qs, l); dts.append(_t.perf_counter() - t0)
        dtc = dts[1]
        print(f"compact steps 2-4 ({label} job), 30 M x 150 under 40 GiB: workspace {engine.workspace_bytes() / 2**30:.1f} GiB, {dtc * 1e3:.0f} ms host arrays to host arrays (first call, incl. the wait for scrubbed memory: {dts[0] * 1e3:.0f} ms)")
        assert engine.workspace_bytes() <= 40 * 2**30
        assert {k: cst[k] for k in ("num_clust", "qs_smoothed", "modified")} == {k: st[k] for k in ("num_clust", "qs_smoothed", "modified")}
        assert np.array_equal(cb, ib) and np.array_equal(cq, iq) and np.array_equal(croff, iroff)
        del cb, cq, croff
    engine.set_params(k=16, m=5, M=2, B=0, v=ord(">"))
    del bwt, qs, lcp
    assert {k: ist[k] for k in ("num_clust", "qs_smoothed", "modified")} == {k: st[k] for k in ("num_clust", "qs_smoothed", "modified")}
    assert torch.equal(torch.from_numpy(ib).to(dev), ob) and torch.equal(torch.from_numpy(iq).to(dev), oq)
    assert int(iroff[-1]) == tot and bool((np.diff(iroff.astype(np.int64)) == L).all())
    del ib, iq, changed_b, changed_q
    ob2 = ob.clone(); oq2 = oq.clone()
    # 2b. step 1 pile by pile (k_piles.hip) at full size: the same reads
    engine.set_params(k=16, m=5, M=2, B=0, v=ord(">"), piles=1)
    stp = run(); torch.cuda.synchronize()
    assert {k: stp[k] for k in st if k != "n_big_segments"} == {k: st[k] for k in st if k != "n_big_segments"}
    assert torch.equal(ob, ob2) and torch.equal(oq, oq2)
    # 2c. the same block under a 40 GiB workspace cap (capped mode: 4.4 n bytes + one two-symbol pile): the same reads
    import time as _time
    engine.set_params(k=16, m=5, M=2, B=0, v=ord(">"), ws_cap_mib=40 * 1024)
    stc = run(); torch.cuda.synchronize()
    t0 = _time.perf_counter(); stc = run(); torch.cuda.synchronize(); dtc = _time.perf_counter() - t0
    print(f"capped mode, 30 M x 150 under 40 GiB: workspace {engine.workspace_bytes() / 2**30:.1f} GiB, {dtc * 1e3:.0f} ms per block")
    assert engine.workspace_bytes() <= 40 * 2**30
    assert {k: stc[k] for k in st if k != "n_big_segments"} == {k: st[k] for k in st if k != "n_big_segments"}
    assert torch.equal(ob, ob2) and torch.equal(oq, oq2)
    # 3. B = 1
    engine.set_params(k=16, m=5, M=2, B=1, v=ord(">"))
    stb = run(); torch.cuda.synchronize()
    assert {k: stb[k] for k in st} == st
    assert torch.equal(ob, ob2)
    lut = torch.tensor([_bin8(c) for c in range(256)], dtype=torch.uint8, device=dev)
    assert torch.equal(oq, lut[oq2.long()])
    # 4. M = 1
    engine.set_params(k=16, m=5, M=1, B=0)
    st1 = run(); torch.cuda.synchronize()
    assert st1["modified"] == st["modified"] and st1["num_clust"] == st["num_clust"] and st1["num_clust_mod"] == st["num_clust_mod"]
    assert torch.equal(ob, ob2)
    ch1 = int((oq != dq).sum().item())
    assert 0 < ch1 <= st1["qs_smoothed"] and st1["bases_inside"] == st["bases_inside"]
    assert not bool(((ob != db) & (oq != dq)).any())                       # a replaced base keeps its quality
    del db, dq, ob, oq, ob2, oq2
    torch.cuda.empty_cache()
    engine.set_params()


def _bin8(c):
    q = (c - 256 if c > 127 else c) - 33
    for lo, v in ((40, 40), (35, 37), (30, 33), (25, 27), (20, 22), (10, 15), (2, 6)):
        if q >= lo:
            q = v
            break
    return (q + 33) & 0xFF


def test_long_and_degenerate_collections(engine, orc):
    """Reads up to 60 000 bases (LCP held in 16 bits), empty collections, only-empty reads, all-N reads,
    poly-A (one segment longer than a wavefront with 3 000-symbol ties); reads beyond BFQ_MAX_READ_LEN refused."""
    rng = np.random.default_rng(5)
    ACGT = np.array(list(b"ACGT"), np.uint8)
    g = ACGT[rng.integers(0, 4, 70000)]
    E = np.zeros(0, np.uint8)

    def mk(reads):
        b = np.concatenate(reads) if sum(len(x) for x in reads) else E
        q = rng.integers(33, 74, len(b)).astype(np.uint8)
        r = np.zeros(len(reads) + 1, np.uint64)
        r[1:] = np.cumsum([len(x) for x in reads])
        return b, q, r

    cases = [[g[:9999], g[:20000], g[:60000], g[100:60000], g[:60000].copy(), g[5:9999]],
             [E] * 5,
             [np.full(50, ord("N"), np.uint8)] * 40 + [np.full(7, ord("N"), np.uint8)],
             [E] * 5000 + [ACGT[:1]],
             [np.full(3000, ord("A"), np.uint8)] * 3 + [np.full(2999, ord("A"), np.uint8)]]
    for reads in cases:
        b, q, r = mk(reads)
        _check_against_oracle(engine, orc, b, q, r, m=2, k=3)
    b, q, r = mk([])                                                   # no reads at all
    hb, hq, st = engine.run_reads(b, q, r)
    assert len(hb) == 0 and st["n_rows"] == 0
    with pytest.raises(api.BfqError) as e:
        engine.run_reads(*mk([g[:65001]]))
    assert e.value.code == -5


def _low_complexity(rng):
    """Poly-A tails, a dinucleotide repeat, noisy poly-G and ordinary reads: ~0.9 M rows, most of them in a few
    equal-prefix segments of 10^5 rows and in clusters just as long."""
    ACGT = np.array(list(b"ACGT"), np.uint8)
    reads = []
    for _ in range(6000):
        reads.append(np.full(int(rng.integers(30, 101)), ord("A"), np.uint8))
    for _ in range(2500):
        L = int(rng.integers(40, 101)); ph = int(rng.integers(0, 2))
        s = np.array([ord("A"), ord("C")], np.uint8)[(np.arange(L) + ph) % 2]
        e = rng.random(L) < 0.01; s[e] = ACGT[rng.integers(0, 4, int(e.sum()))]
        reads.append(s)
    for _ in range(2500):
        L = int(rng.integers(20, 101))
        s = np.full(L, ord("G"), np.uint8)
        e = rng.random(L) < 0.02; s[e] = ACGT[rng.integers(0, 4, int(e.sum()))]
        s[rng.random(L) < 0.005] = ord("N")
        reads.append(s)
    g = ACGT[rng.integers(0, 4, 3000)]
    for _ in range(2000):
        st = int(rng.integers(0, 2900)); s = g[st:st + 100].copy()
        if rng.random() < 0.3:
            s[int(rng.integers(30, 80)):] = ord("A")
        reads.append(s)
    reads = [reads[i] for i in rng.permutation(len(reads))]
    b = np.concatenate(reads)
    q = rng.integers(33, 74, len(b)).astype(np.uint8)
    r = np.zeros(len(reads) + 1, np.uint64)
    r[1:] = np.cumsum([len(x) for x in reads])
    return b, q, r


def test_low_complexity_collections(engine, orc, monkeypatch):
    """Segments and clusters of 10^5 rows (k_bigseg.hip's radix rounds, k_cluster_big) in every mode; then the same
    with the rounds' slot budget cut down (BFQ_HUGE_CAP) so that batching and the one-workgroup route for
    segments beyond the budget run too."""
    b, q, r = _low_complexity(np.random.default_rng(7))
    for M in range(4):
        st = _check_against_oracle(engine, orc, b, q, r, M=M, B=M & 1, m=5)
    assert st["n_big_segments"] > 0
    monkeypatch.setenv("BFQ_HUGE_CAP", "60000")
    _check_against_oracle(engine, orc, b, q, r, m=2, k=20)


def test_long_cluster_two_frequent_bases(engine, orc):
    """A cluster of 6 000 rows whose eBWT symbols are C and G half and half, each with its own preceding base: the
    two-frequent-symbol branch (bfq_int.cpp:542-591) inside the tiled long-cluster kernels, incl. base replacement."""
    rng = np.random.default_rng(11)
    ACGT = np.array(list(b"ACGT"), np.uint8)
    U = ACGT[rng.integers(0, 4, 60)]
    mk = lambda pre: np.concatenate([np.frombuffer(pre, np.uint8), U])
    reads = [mk(b"AC")] * 3000 + [mk(b"TG")] * 3000 + [mk(b"AT")] * 40 + [mk(b"TA")] * 40
    reads = [reads[i] for i in rng.permutation(len(reads))]
    b = np.concatenate(reads)
    q = rng.integers(33, 74, len(b)).astype(np.uint8)
    r = np.zeros(len(reads) + 1, np.uint64)
    r[1:] = np.cumsum([len(x) for x in reads])
    for i, x in enumerate(reads):                                      # the odd bases before U are never trusted (-t 20)
        if bytes(x[:2]) in (b"AT", b"TA"):
            q[int(r[i]) + 1] = 35
    for M in (2, 0, 1, 3):
        st = _check_against_oracle(engine, orc, b, q, r, M=M, m=5)
        assert st["num_clust_mod"] > 0 and st["modified"] > 0


def test_randomised_shapes_and_parameters(engine, orc):
    """The first 120 cases of tests/soak_gpu.py (random collection shapes x random parameters)."""
    from tests import soak_gpu
    for seed in range(100000, 100120):
        ok, rows, what = soak_gpu.run_case(engine, orc, seed)
        assert ok, what
