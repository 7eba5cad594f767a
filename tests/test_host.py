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


# ---- per-GPU lease of the one-shot tools (bfq_device_lease: host only, no GPU needed).  BFQzip_parallel.py:277-285
# starts n concurrent children; each must end up on a GPU of its own, and never two on one.
_LEASE_CHILD = r"""
import ctypes as C, os, sys, time
L = C.CDLL({lib!r})                     # no torch import: the children must start within milliseconds of each other
L.bfq_device_lease.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double)]
path = C.create_string_buffer(300); waited = C.c_double(0)
ids = (C.c_char_p * {n})(*[("t%d" % k).encode() for k in range({n})])
t0 = time.time()
slot = L.bfq_device_lease({n}, ids, {only}, path, 300, C.byref(waited))
t1 = time.time()
print("GOT", slot, path.value.decode(), "%.3f" % t0, "%.3f" % t1, "%.3f" % waited.value, flush=True)
time.sleep({hold})
if {release}:
    assert L.bfq_device_release(slot) == 0
    print("REL", "%.3f" % time.time(), flush=True)
    time.sleep(0.3)
print("END", "%.3f" % time.time(), flush=True)
"""


def _lease_children(tmp_path, n_slots, n_children, hold, only=-1, release=0, stagger=0.0):
    import subprocess, sys, time
    env = dict(os.environ, BFQ_LEASE_DIR=str(tmp_path))
    code = _LEASE_CHILD.format(lib=_lib.LIB_PATH, n=n_slots, only=only, hold=hold, release=release)
    ps = []
    for _ in range(n_children):
        ps.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True))
        if stagger:
            time.sleep(stagger)
    outs = []
    for p in ps:
        o, _ = p.communicate(timeout=120)
        assert p.returncode == 0, o
        rec = {}
        for line in o.splitlines():
            f = line.split()
            if f and f[0] == "GOT":
                rec.update(slot=int(f[1]), path=f[2], t_ask=float(f[3]), t_got=float(f[4]), waited=float(f[5]))
            elif f and f[0] in ("REL", "END"):
                rec[f[0].lower()] = float(f[1])
        outs.append(rec)
    return outs


def test_device_lease_spreads_concurrent_tools(tmp_path):
    """Two concurrent holders with two slots take different slots and neither waits."""
    r = _lease_children(tmp_path, 2, 2, hold=1.5)
    assert sorted(x["slot"] for x in r) == [0, 1]
    assert len({x["path"] for x in r}) == 2 and all(str(tmp_path) in x["path"] for x in r)
    assert all(x["waited"] < 1.0 for x in r)


def test_device_lease_serialises_on_one_slot(tmp_path):
    """With a single slot the second tool waits until the first process has ended (the kernel drops the lock with it)."""
    r = _lease_children(tmp_path, 1, 2, hold=1.0, stagger=0.3)
    assert [x["slot"] for x in r] == [0, 0]
    first, second = sorted(r, key=lambda x: x["t_got"])
    assert second["t_got"] >= first["end"] - 0.05 and second["waited"] > 0.3


def test_device_lease_more_tools_than_slots_and_release(tmp_path):
    """Three tools, two slots: the third gets whichever slot is given back first (bfq_device_release), not a fixed one."""
    r = _lease_children(tmp_path, 2, 3, hold=0.8, release=1, stagger=0.15)
    got = sorted(r, key=lambda x: x["t_got"])
    assert sorted(x["slot"] for x in got[:2]) == [0, 1]
    assert got[2]["waited"] > 0.2 and got[2]["t_got"] >= min(got[0]["rel"], got[1]["rel"]) - 0.05


def test_device_lease_pinned_slot(tmp_path):
    """only_slot (BFQ_DEVICE): both tools ask for slot 1 of 4 and run one after the other on it."""
    r = _lease_children(tmp_path, 4, 2, hold=0.7, only=1, stagger=0.2)
    assert [x["slot"] for x in r] == [1, 1]
    first, second = sorted(r, key=lambda x: x["t_got"])
    assert second["t_got"] >= first["end"] - 0.05
    L = _lib.lib()
    assert L.bfq_device_lease(0, None, -1, None, 0, None) == -1 and L.bfq_device_lease(2, None, 2, None, 0, None) == -1


def test_phase_report_line(tmp_path):
    """The tools' -V / BFQ_TRACE timeline: one JSON object per process on stderr, phases in order of first use."""
    import json, subprocess, sys
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from bfqzip_amd import _lib\nL = _lib.lib()\nL.bfq_phase_enable(1)\n"
            "L.bfq_phase(b'start'); time.sleep(0.05); L.bfq_phase(b'gpu'); time.sleep(0.1); L.bfq_phase(b'start'); time.sleep(0.05)\n"
            "L.bfq_phase_report(b'tool')\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code], stderr=subprocess.PIPE, text=True, timeout=120)
    line = [x for x in p.stderr.splitlines() if x.startswith("[bfq phases] ")]
    assert len(line) == 1, p.stderr
    d = json.loads(line[0][len("[bfq phases] "):])
    assert list(d)[:2] == ["tool", "exec_to_main"] and list(d)[2:] == ["start", "gpu", "total"]
    assert 0.08 <= d["start"] <= 0.5 and 0.08 <= d["gpu"] <= 0.5 and d["total"] >= d["start"] + d["gpu"] and d["exec_to_main"] >= 0


def test_file_put_places_ranges_of_one_file(tmp_path):
    """bfq_file_put: every rank writes its byte range of a shared output file (bfqzip_amd/parallel.py; the reference merges
    with `cat`, BFQzip_parallel.py:174-177).  Ranges in any order, unaligned offsets, several threads, a file that grows
    and is never shrunk; a descriptor that cannot be mapped falls back to pwrite."""
    rng = np.random.default_rng(1)
    data = rng.integers(0, 256, 9_000_001, dtype=np.uint8)
    cuts = [0, 3, 4097, 1_000_000, 5_000_001, len(data)]
    path = str(tmp_path / "out.bin")
    open(path, "wb").close()
    order = [3, 0, 4, 2, 1]
    for k in order:                                                   # the last range first: the file grows, nothing already there is lost
        fd = os.open(path, os.O_RDWR)
        api.file_put(fd, data[cuts[k]:cuts[k + 1]], cuts[k], threads=1 + k % 3)
        os.close(fd)
    assert np.array_equal(np.fromfile(path, np.uint8), data)
    api.file_put(os.open(path, os.O_RDWR), data[:0], 0)               # nothing to write
    # two processes at once, disjoint halves
    import subprocess, sys
    half = len(data) // 2
    np.save(str(tmp_path / "d.npy"), data)
    p2 = str(tmp_path / "two.bin")
    open(p2, "wb").close()
    code = ("import sys, os, numpy as np; sys.path.insert(0, %r)\nfrom bfqzip_amd import api\n"
            "d = np.load(sys.argv[1]); a, b = int(sys.argv[3]), int(sys.argv[4])\n"
            "fd = os.open(sys.argv[2], os.O_RDWR); api.file_put(fd, d[a:b], a); os.close(fd)\n") % ROOT
    ps = [subprocess.Popen([sys.executable, "-c", code, str(tmp_path / "d.npy"), p2, str(a), str(b)]) for a, b in ((half, len(data)), (0, half))]
    assert all(p.wait(timeout=300) == 0 for p in ps)
    assert np.array_equal(np.fromfile(p2, np.uint8), data)
    env = dict(os.environ, BFQ_NO_OUTMAP="1")                         # the pwrite route
    p3 = str(tmp_path / "three.bin")
    open(p3, "wb").close()
    assert subprocess.run([sys.executable, "-c", code, str(tmp_path / "d.npy"), p3, "0", str(len(data))], env=env, timeout=300).returncode == 0
    assert np.array_equal(np.fromfile(p3, np.uint8), data)


def test_file_range_is_the_files_pages(tmp_path):
    """bfq_file_map / FileRange: a byte range of an output file as memory (allocated, mapped, populated): what the global
    mode hands to the GPU transfers as their destination, so that nothing has to be written afterwards."""
    path = str(tmp_path / "r.bin")
    open(path, "wb").close()
    fd = os.open(path, os.O_RDWR)
    a = api.FileRange(fd, 5000, 3_000_000)                      # unaligned start, beyond the current end of the file
    b = api.FileRange(fd, 0, 5000)
    assert a.array is not None and len(a.array) == 3_000_000 and b.array is not None
    rng = np.random.default_rng(2)
    da = rng.integers(0, 256, len(a.array), dtype=np.uint8); db = rng.integers(0, 256, 5000, dtype=np.uint8)
    a.array[:] = da; b.array[:] = db
    a.close(); b.close()
    os.close(fd)
    got = np.fromfile(path, np.uint8)
    assert len(got) == 3_005_000 and np.array_equal(got[:5000], db) and np.array_equal(got[5000:], da)
    assert api.FileRange(os.open(path, os.O_RDWR), 0, 0).array is None      # nothing to map
