"""CPU tests of the host side: the C-ABI library loads and exports every symbol that
include/bfqzip_hip.h declares (no compute without a GPU), FASTQ helpers, the seeded
generator, and loud failure when no GPU is present."""
import ctypes as C
import os, re
import numpy as np
import pytest
from bfqzip_amd import _lib, api, fastq
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bfqzip_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bfq_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/bfqzip_hip.h but not exported"
    assert set(_lib.SYMBOLS) <= declared
    assert b"gfx950" in L.bfq_version()


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.Params) == 16 * 4
    assert C.sizeof(_lib.Stats) == 12 * 8
    assert C.sizeof(_lib.Synth) == 8 + 8 + 13 * 4 + 4   # padded to 8
    p = _lib.Params()
    _lib.lib().bfq_default_params(C.byref(p))
    assert (p.K, p.m, p.v, p.f, p.t, p.term, p.M, p.B) == (16, 2, ord(">"), 40, 20, ord("#"), 2, 0)   # bfq_int.cpp:70-90


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.BfqError) as e:
        api.Engine(0)
    assert "no CPU fallback" in str(e.value)


def test_fastq_roundtrip_and_validation():
    raw = open(os.path.join(util.GOLDEN, "example.fastq"), "rb").read()
    b, q, r, h = fastq.parse_fastq_bytes(raw)
    assert len(r) == 101 and int(r[-1]) == len(b) == len(q) == 10100
    assert fastq.format_fastq(b, q, r, h) == raw
    assert fastq.format_fastq(b[:0], q[:0], r[:1]) == b""
    with pytest.raises(ValueError):
        fastq.parse_fastq_bytes(b"@r\nACGT\n+\nIII\n")           # checkFASTQ.py:18-32
    with pytest.raises(ValueError):
        fastq.parse_fastq_bytes(b"@r\nACGT\n+\n")
    b2, q2, r2, _ = fastq.parse_fastq_bytes(b"@a\nAC\n+\nII\n@b\n\n+\n\n@c\nG\n+\nI")   # empty read, no final newline
    assert list(r2) == [0, 2, 2, 3] and b2.tobytes() == b"ACG"
    lines = raw.split(b"\n")
    assert fastq.format_lines(b, r) == b"".join(x + b"\n" for x in lines[1::4])      # sed -n 2~4p (BFQzip.py:21)
    assert fastq.format_lines(q, r) == b"".join(x + b"\n" for x in lines[3::4])
    assert fastq.format_lines(b2, r2) == b"AC\n\nG\n" and fastq.format_lines(b[:0], r[:1]) == b""


def test_synthetic_generator_is_seeded_and_shaped():
    sp = api.synth_spec(2000, 100, seed=5)
    b, q, r = api.synth_host(sp)
    b2, q2, r2 = api.synth_host(api.synth_spec(2000, 100, seed=5))
    assert np.array_equal(b, b2) and np.array_equal(q, q2) and np.array_equal(r, r2)
    b3, _, _ = api.synth_host(api.synth_spec(2000, 100, seed=6))
    assert not np.array_equal(b, b3)
    assert set(np.unique(b)) <= set(b"ACGTN") and q.min() >= 33 and q.max() <= 33 + 41
    assert list(np.diff(r.astype(np.int64))) == [100] * 2000
    # reads are prefixes of the same generator whatever N' >= N is asked for... per read independence
    sp2 = api.synth_spec(2000, 100, seed=5); sp2.coverage = 30
    frac_n = (b == ord("N")).mean()
    assert 0.0002 < frac_n < 0.003
    spv = api.synth_spec(500, 20, Lmax=70, seed=9)
    bv, qv, rv = api.synth_host(spv)
    lens = np.diff(rv.astype(np.int64))
    assert lens.min() >= 20 and lens.max() <= 70 and len(bv) == int(rv[-1])


def test_shared_host_device_helpers(tmp_path):
    """bfq_common.h (key windows, terminator masking, key LCP, sort records, binning) compiled for the host."""
    import subprocess
    exe = str(tmp_path / "test_common")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cxx", "test_common.cpp")])
    r = subprocess.run([exe], stdout=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
