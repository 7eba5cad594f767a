"""CPU tests: the oracle (oracle/bfq_oracle.c) against the golden vectors made by the
compiled reference (tests/golden/make_golden.py), and against the reference
binary itself (oracle/_ref, when present) on random small inputs."""
import os, subprocess, tempfile
import numpy as np
import pytest
from bfqzip_amd import fastq
from tests import util

IDX = util.golden_index()
CASES = [(name, key) for name in IDX for key in IDX[name]["out"]]


@pytest.mark.parametrize("name", list(IDX))
def test_oracle_ebwt_matches_golden(orc, name):
    b, q, r, h, bwt, qs, lcp = util.golden_set(name)
    bwt2, qs2, lcp2 = orc.build_ebwt(b, q, r)
    assert util.md5(bwt2.tobytes()) == IDX[name]["bwt_md5"]
    assert util.md5(qs2.tobytes()) == IDX[name]["qs_md5"]
    assert np.array_equal(lcp2.astype(np.uint16), lcp)
    assert len(bwt2) == IDX[name]["n"]


@pytest.mark.parametrize("name,key", CASES)
def test_oracle_matches_reference_output(orc, name, key):
    b, q, r, h, bwt, qs, lcp = util.golden_set(name)
    d, hdr = util.parse_case(key)
    p = orc.params(K=d["k"], m=d["m"], v=d["v"], f=d["f"], t=d["t"], M=d["M"], B=d["B"])
    ob, oq, oroff, st = orc.smooth_invert(bwt, qs, None, p)           # LCP deduced from the BWT
    out = fastq.format_fastq(ob, oq, oroff, h if hdr else None)
    assert util.md5(out) == IDX[name]["out"][key]
    ob2, oq2, oroff2, st2 = orc.smooth_invert(bwt, qs, lcp.astype(np.uint32), p)   # explicit LCP (bfq_ext mode)
    assert np.array_equal(ob, ob2) and np.array_equal(oq, oq2) and st == st2
    if key == "M2B0 -m 5":
        assert out == open(os.path.join(util.GOLDEN, name + ".M2B0.fq"), "rb").read()


def test_example_known_answers(orc):
    """SURVEY Appendix B facts for example/reads.fastq, M=2 B=0 -m 5."""
    b, q, r, h, bwt, qs, lcp = util.golden_set("example")
    assert bwt[:40].tobytes() == b"CCTGGAAAGAGGGTGCGGCGCCCCCCTATGATAACACTGT"
    ob, oq, oroff, st = orc.smooth_invert(bwt, qs, None, orc.params(m=5))
    assert (st["num_clust"], st["bases_inside"], st["qs_smoothed"], st["modified"]) == (387, 4210, 4198, 10)
    ob, oq, oroff, st = orc.smooth_invert(bwt, qs, None, orc.params(K=10000))
    assert np.array_equal(ob, b) and np.array_equal(oq, q) and st["num_clust"] == 0


def test_oracle_vs_reference_binary_fuzz(orc):
    """Differential fuzz against oracle/_ref/bfq_int_M?_B? (the reference compiled here)."""
    if orc.ref_binary(2, 0) is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(7)
    with tempfile.TemporaryDirectory() as d:
        for it in range(60):
            nreads = int(rng.integers(1, 120)); lmax = int(rng.integers(1, 50))
            b, q, r = util.random_reads(rng, nreads, 1, lmax)
            M, B = int(rng.integers(0, 4)), int(rng.integers(0, 2))
            K = int(rng.choice([1, 2, 3, 5, 8, 16])); m = int(rng.choice([2, 3, 5, 9]))
            v = int(rng.choice([62, 53, 73])); t = int(rng.choice([5, 20, 35])); f = int(rng.choice([40, 50, 70]))
            bwt, qs, lcp = orc.build_ebwt(b, q, r)
            bwt.tofile(d + "/x.bwt"); qs.tofile(d + "/x.bwt.qs")
            cmd = [orc.ref_binary(M, B), "-e", d + "/x.bwt", "-q", d + "/x.bwt.qs", "-o", d + "/o.fq",
                   "-k", str(K), "-m", str(m), "-v", str(v), "-t", str(t), "-f", str(f)]
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, timeout=60)
            ref = open(d + "/o.fq", "rb").read()
            p = orc.params(K=K, m=m, v=v, f=f, t=t, M=M, B=B)
            ob, oq, oroff, st = orc.smooth_invert(bwt, qs, None, p)
            assert fastq.format_fastq(ob, oq, oroff) == ref, (it, M, B, K, m, v, t, f)


def test_tie_order_of_identical_suffixes_is_free(orc, tmp_path):
    """The reference sees ONE terminator symbol: an eBWT whose identical suffixes (and terminator rows) are in any
    order is processed all the same, LCP by the usual convention.  Pins, against the compiled reference, the contract the
    GPU's BWT-only LCP deduction (k_bfs.hip) is tested with in tests/test_gpu_parity.py."""
    import subprocess
    ref = orc.ref_binary(2, 0)
    if ref is None:
        pytest.skip("reference bfq_int not built (oracle/_ref)")
    rng = np.random.default_rng(99)
    done = 0
    for it in range(12):
        b, q, r = util.random_reads(rng, int(rng.integers(5, 60)), 1, int(rng.integers(4, 40)), dup=0.45, p_n=0.02)
        bwt, qs, lcp = orc.build_ebwt(b, q, r)
        sb, sq = util.shuffle_ties(bwt, qs, rng)
        if np.array_equal(sb, bwt):
            continue
        suf, reads = util.decode_rows(sb)
        lc = util.lcp_of_rows(suf)
        p = orc.params(m=2, K=int(rng.choice([2, 3, 5])))
        ob, oq, oroff, st = orc.smooth_invert(sb, sq, lc, p)
        ob2, oq2, oroff2, st2 = orc.smooth_invert(sb, sq, None, p)            # the oracle's own BWT-only deduction
        assert np.array_equal(ob, ob2) and np.array_equal(oq, oq2) and st == st2
        sb.tofile(str(tmp_path / "x.bwt")); sq.tofile(str(tmp_path / "x.bwt.qs"))
        subprocess.check_call([ref, "-e", str(tmp_path / "x.bwt"), "-q", str(tmp_path / "x.bwt.qs"), "-o", str(tmp_path / "o.fq"),
                               "-m", "2", "-k", str(p.K)], stdout=subprocess.DEVNULL)
        assert open(str(tmp_path / "o.fq"), "rb").read() == fastq.format_fastq(ob, oq, oroff)
        done += 1
    assert done >= 6
