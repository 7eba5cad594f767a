"""GPU tests of the drop-in front-ends: the argv the reference drivers produce
(BFQzip.py:184,215-222; BFQzip_ext.py:172-177,208-214 -- SURVEY.md 8(b)) must yield
the files the reference tools would."""
import os, shutil, subprocess
import numpy as np
import pytest
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROP = os.path.join(ROOT, "dropin")
IDX = util.golden_index()


def _run(cmd, **kw):
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, **kw)


@pytest.fixture(scope="module")
def tools():
    t = {k: os.path.join(DROP, p) for k, p in dict(gsufsort="external/gsufsort/gsufsort", egap="external/egap/eGap",
                                                   bfq_int="src_int_mem/bfq_int", bfq_ext="src_ext_mem/bfq_ext").items()}
    if not all(os.path.exists(p) for p in t.values()):          # normally prebuilt by __graft_entry__.build()
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bfqzip_amd", "csrc"), "cli"])
    for p in t.values():
        assert os.path.exists(p), f"{p} missing: run __graft_entry__.build()"
    return t


@pytest.mark.parametrize("name", ["example", "synth_var"])
def test_int_mem_pipeline(tools, tmp_path, name):
    fq = os.path.join(util.GOLDEN, name + ".fastq")
    out = str(tmp_path / "OUT")
    r = _run([tools["gsufsort"], fq, "--bwt", "--qs", "-o", out])
    assert r.returncode == 0, r.stdout
    assert util.md5(open(out + ".bwt", "rb").read()) == IDX[name]["bwt_md5"]
    assert util.md5(open(out + ".bwt.qs", "rb").read()) == IDX[name]["qs_md5"]
    r = _run([tools["bfq_int"], "-e", out + ".bwt", "-q", out + ".bwt.qs", "-o", out + ".fq", "-m", "5"])
    assert r.returncode == 0, r.stdout
    assert open(out + ".fq", "rb").read() == open(os.path.join(util.GOLDEN, name + ".M2B0.fq"), "rb").read()
    # other (M,B) builds of the reference = BFQ_M / BFQ_B here; -k/-v/-t/-f flags as the driver passes them
    for key, want in IDX[name]["out"].items():
        d, hdr = util.parse_case(key)
        if hdr:
            continue
        cmd = [tools["bfq_int"], "-e", out + ".bwt", "-q", out + ".bwt.qs", "-o", out + ".x.fq", "-m", str(d["m"]),
               "-k", str(d["k"]), "-v", str(d["v"]), "-t", str(d["t"]), "-f", str(d["f"])]
        r = _run(cmd, env=dict(os.environ, BFQ_M=str(d["M"]), BFQ_B=str(d["B"])))
        assert r.returncode == 0, r.stdout
        assert util.md5(open(out + ".x.fq", "rb").read()) == want, key


def test_headers_option(tools, tmp_path):
    name = "example"
    fq = os.path.join(util.GOLDEN, name + ".fastq")
    out = str(tmp_path / "OUT")
    with open(out + ".h", "wb") as f:                                  # sed -n 1~4p (BFQzip.py:192-203)
        f.write(b"".join(l for i, l in enumerate(open(fq, "rb").readlines()) if i % 4 == 0))
    assert _run([tools["gsufsort"], fq, "--bwt", "--qs", "-o", out]).returncode == 0
    r = _run([tools["bfq_int"], "-e", out + ".bwt", "-q", out + ".bwt.qs", "-o", out + ".fq", "-m", "5", "-H", out + ".h"])
    assert r.returncode == 0, r.stdout
    assert util.md5(open(out + ".fq", "rb").read()) == IDX[name]["out"]["M2B0 -m 5 -H"]


def test_ext_mem_pipeline(tools, tmp_path):
    name = "synth_fix"
    fq = os.path.join(util.GOLDEN, name + ".fastq")
    out = str(tmp_path / "OUT")
    r = _run([tools["egap"], fq, "--em", "--mem", "4096", "--qs", "-o", out, "--lcp", "--lbytes", "1"])
    assert r.returncode == 0, r.stdout
    bwt = np.fromfile(out + ".bwt", np.uint8)
    gold = np.fromfile(os.path.join(util.GOLDEN, name + ".bwt"), np.uint8)
    assert np.array_equal(np.where(bwt == 0, ord("#"), bwt), gold)     # terminator byte 0 (BFQzip_ext.py:208 -s 0)
    lcp = np.fromfile(out + ".1.lcp", np.uint8)
    assert np.array_equal(lcp, np.minimum(np.fromfile(os.path.join(util.GOLDEN, name + ".lcp16"), np.uint16), 255))
    r = _run([tools["bfq_ext"], "-e", out + ".bwt", "-q", out + ".bwt.qs", "-a", out + ".1.lcp", "-o", out, "-l", "250",
              "-s", "0", "-m", "5"])
    assert r.returncode == 0, r.stdout
    assert open(out + ".fq", "rb").read() == open(os.path.join(util.GOLDEN, name + ".M2B0.fq"), "rb").read()


def test_error_behaviour(tools, tmp_path):
    # usage / missing file: help and exit 0 (bfq_int.cpp:132,937-955)
    r = _run([tools["bfq_int"], "-e", "/nonexistent.bwt", "-q", "/nonexistent.qs", "-o", str(tmp_path / "o.fq")])
    assert r.returncode == 0 and b"could not find file" in r.stdout
    # forbidden eBWT symbol: exit 1 (dna_string_n.hpp:87-93)
    (tmp_path / "b.bwt").write_bytes(b"AXC#"); (tmp_path / "b.qs").write_bytes(b"IIII")
    r = _run([tools["bfq_int"], "-e", str(tmp_path / "b.bwt"), "-q", str(tmp_path / "b.qs"), "-o", str(tmp_path / "o.fq")])
    assert r.returncode == 1
    # builder failure -> non-zero (the driver exits 1, BFQzip.py:99-100)
    (tmp_path / "bad.fastq").write_bytes(b"@r\nACGT\n+\nII\n")
    assert _run([tools["gsufsort"], str(tmp_path / "bad.fastq"), "--bwt", "--qs", "-o", str(tmp_path / "X")]).returncode == 1
