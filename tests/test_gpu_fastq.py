"""GPU tests of the FASTQ text entry points: parsing and formatting on the device
must give the files the reference pipeline gives (goldens written by the compiled reference bfq_int)."""
import os
import numpy as np
import pytest
from bfqzip_amd import api, fastq
from tests import util

pytestmark = pytest.mark.gpu
IDX = util.golden_index()


def _raw(name):
    return open(os.path.join(util.GOLDEN, name + ".fastq"), "rb").read()


@pytest.mark.parametrize("name", list(IDX))
def test_fastq_build_ebwt(engine, name):
    b, q, r, h, bwt, qs, lcp = util.golden_set(name)
    gb, gq, gl = engine.fastq_build_ebwt(_raw(name))
    assert np.array_equal(gb, bwt) and np.array_equal(gq, qs) and np.array_equal(gl, lcp)
    eb, eq, _ = engine.fastq_build_ebwt(_raw(name), term_out=0, want_lcp=False)     # eGap: terminator byte 0
    assert np.array_equal(np.where(eb == 0, ord("#"), eb), bwt)


@pytest.mark.parametrize("name", list(IDX))
def test_fastq_run_text_to_text(engine, name):
    engine.set_params(m=5)
    out, st = engine.fastq_run(_raw(name))
    assert out == open(os.path.join(util.GOLDEN, name + ".M2B0.fq"), "rb").read()
    key = "M2B0 -m 5 -H"
    if key in IDX[name]["out"]:
        outh, sth = engine.fastq_run(_raw(name), keep_headers=True)
        assert util.md5(outh) == IDX[name]["out"][key]
        assert sth == st


@pytest.mark.parametrize("name", list(IDX))
def test_fastq_run_streams(engine, name):
    """OUT.fq.dna / OUT.fq.qs / OUT.h of BFQzip.py --m2/--m3 = `sed -n 2~4p / 4~4p` of the reference's OUT.fq
    (golden, written by the compiled bfq_int) and `sed -n 1~4p` of the input (BFQzip.py:19-21,192-251)."""
    engine.set_params(m=5)
    dna, qs, hdr, st = engine.fastq_run_streams(_raw(name))
    ref = open(os.path.join(util.GOLDEN, name + ".M2B0.fq"), "rb").read().split(b"\n")[:-1]
    assert dna == b"".join(x + b"\n" for x in ref[1::4])
    assert qs == b"".join(x + b"\n" for x in ref[3::4])
    lines = _raw(name).split(b"\n")
    if lines[-1] == b"":
        lines.pop()
    assert hdr == b"".join(x + b"\n" for x in lines[0::4])
    d2, q2, h2, st2 = engine.fastq_run_streams(_raw(name), want_headers=False)
    assert h2 is None and d2 == dna and q2 == qs and st2 == st
    d0, q0, h0, _ = engine.fastq_run_streams(b"")
    assert d0 == b"" and q0 == b"" and h0 == b""


def test_smooth_invert_fastq_with_header_file(engine):
    name = "example"
    b, q, r, h, bwt, qs, lcp = util.golden_set(name)
    engine.set_params(m=5)
    out, st = engine.smooth_invert_fastq(bwt, qs)
    assert out == open(os.path.join(util.GOLDEN, name + ".M2B0.fq"), "rb").read()
    hdr = b"".join(x + b"\n" for x in h)                                    # sed -n 1~4p
    outh, _ = engine.smooth_invert_fastq(bwt, qs, lcp, headers=hdr)
    assert util.md5(outh) == IDX[name]["out"]["M2B0 -m 5 -H"]
    outn, _ = engine.smooth_invert_fastq(bwt, qs, lcp, headers=hdr[:-1])    # last header line without newline
    assert outn == outh
    with pytest.raises(api.BfqError):
        engine.smooth_invert_fastq(bwt, qs, lcp, headers=hdr[:200])        # too few header lines


def test_fastq_text_edge_cases(engine, orc):
    engine.set_params(m=2, k=1)
    cases = [b"@a\nACGT\n+\nIIII\n@b\nAC\n+x\nI#\n",
             b"@a\r\nACGT\r\n+\r\nIIII\r\n@b\r\nAC\r\n+\r\nI#",          # CRLF, no final newline
             b"@only\nA\n+\nI",
             b"@e\n\n+\n\n@f\nGATTACA\n+\nIIIIIII\n",                      # an empty read
             b""]
    for raw in cases:
        b, q, r, h = fastq.parse_fastq_bytes(raw.replace(b"\r", b""))
        ob, oq, st = orc.run_reads(b, q, r, orc.params(m=2, K=1))
        out, gst = engine.fastq_run(raw)
        assert out == fastq.format_fastq(ob, oq, r), raw
        gb, gq, gl = engine.fastq_build_ebwt(raw)
        eb, eq, el = orc.build_ebwt(b, q, r)
        assert np.array_equal(gb, eb) and np.array_equal(gq, eq) and np.array_equal(gl.astype(np.uint32), el)
    for bad in (b"@a\nACGT\n+\nIII\n", b"@a\nACGT\n+\n", b"@a\nACXT\n+\nIIII\n"):
        with pytest.raises(api.BfqError):
            engine.fastq_run(bad)


def test_fastq_run_large_random(engine, orc):
    sp = api.synth_spec(20000, 30, Lmax=120, seed=99)
    b, q, r = api.synth_host(sp)
    hdrs = [b"@read%d some text" % i for i in range(len(r) - 1)]
    raw = fastq.format_fastq(b, q, r, hdrs)
    engine.set_params(m=5, M=1, B=1)
    ob, oq, st = orc.run_reads(b, q, r, orc.params(m=5, M=1, B=1))
    out, gst = engine.fastq_run(raw, keep_headers=True)
    assert out == fastq.format_fastq(ob, oq, r, hdrs)
    for k in st:
        assert st[k] == gst[k]
