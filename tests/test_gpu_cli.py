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


def _synth_file(engine, path, reads, L, seed=5):
    from bfqzip_amd import api
    buf = np.zeros(reads * (2 * L + 30) + 4096, np.uint8)
    ln = engine.synth_fastq(api.synth_spec(reads, L, seed=seed), buf)
    buf[:ln].tofile(path)
    return buf[:ln]


def test_tools_on_files_beyond_one_staging_chunk(tools, tmp_path, engine):
    """The one-shot paths of the tools (outputs mapped, pre-faulted and written pile by pile by background writers;
    bfq_int's quality upload running beside the LCP deduction) on files of several 16 MiB staging chunks: the same bytes as
    the in-process calls give -- gsufsort / eGap (--lbytes 1 and 4) / bfq_int (with and without -H) / bfq_ext."""
    fq = str(tmp_path / "in.fastq")
    text = _synth_file(engine, fq, 700_000, 100)
    engine.set_params(m=5)
    bwt, qs, lcp = engine.fastq_build_ebwt(text)
    out = str(tmp_path / "OUT")
    env = dict(os.environ, BFQ_TRACE="1")
    r = _run([tools["gsufsort"], fq, "--bwt", "--qs", "-o", out], env=env)
    assert r.returncode == 0 and b"[bfq phases]" in r.stdout, r.stdout
    assert np.array_equal(np.fromfile(out + ".bwt", np.uint8), bwt) and np.array_equal(np.fromfile(out + ".bwt.qs", np.uint8), qs)
    with open(out + ".h", "wb") as f:
        f.write(b"".join(l for i, l in enumerate(open(fq, "rb").readlines()) if i % 4 == 0))
    hdr = np.fromfile(out + ".h", np.uint8)
    for extra, want in (([], engine.smooth_invert_fastq(bwt, qs)[0]), (["-H", out + ".h"], engine.smooth_invert_fastq(bwt, qs, headers=hdr)[0])):
        r = _run([tools["bfq_int"], "-e", out + ".bwt", "-q", out + ".bwt.qs", "-o", out + ".fq", "-m", "5", "-V"] + extra)
        assert r.returncode == 0 and b"[bfq phases]" in r.stdout, r.stdout
        assert np.fromfile(out + ".fq", np.uint8).tobytes() == want
    for lb in (1, 4):
        r = _run([tools["egap"], fq, "--em", "--mem", "4096", "--qs", "-o", out + "e", "--lcp", "--lbytes", str(lb)])
        assert r.returncode == 0, r.stdout
        eb = np.fromfile(out + "e.bwt", np.uint8)
        assert np.array_equal(np.where(eb == 0, ord("#"), eb), bwt) and np.array_equal(np.fromfile(out + "e.bwt.qs", np.uint8), qs)
        got = np.fromfile(out + f"e.{lb}.lcp", np.uint8 if lb == 1 else np.uint32)
        assert np.array_equal(got, np.minimum(lcp, 255) if lb == 1 else lcp.astype(np.uint32))
    r = _run([tools["bfq_ext"], "-e", out + "e.bwt", "-q", out + "e.bwt.qs", "-a", out + "e.4.lcp", "-o", out + "x", "-l", "250", "-s", "0", "-m", "5"])
    assert r.returncode == 0, r.stdout
    engine.set_params(m=5, s=0, ext=1)
    want = engine.smooth_invert_fastq(np.fromfile(out + "e.bwt", np.uint8), qs, lcp=lcp.astype(np.uint32))[0]
    engine.set_params()
    assert np.fromfile(out + "x.fq", np.uint8).tobytes() == want
    # the same tools under a workspace cap (BFQ_WS_CAP): steps 2-4 without the LF table (k_compact.hip), same files
    capenv = dict(os.environ, BFQ_WS_CAP="1100M", BFQ_COMPACT_WIN="4M", BFQ_TRACE="1")
    engine.set_params(m=5)
    want_int = engine.smooth_invert_fastq(bwt, qs)[0]
    engine.set_params()
    r = _run([tools["bfq_int"], "-e", out + ".bwt", "-q", out + ".bwt.qs", "-o", out + ".cap.fq", "-m", "5"], env=capenv)
    assert r.returncode == 0 and np.fromfile(out + ".cap.fq", np.uint8).tobytes() == want_int, r.stdout[-600:]
    ws = [float(x) for x in __import__("re").findall(rb"workspace ([0-9.]+) GiB", r.stdout)]
    assert ws and max(ws) <= 1100 / 1024 + 0.01, r.stdout[-600:]
    r = _run([tools["bfq_ext"], "-e", out + "e.bwt", "-q", out + "e.bwt.qs", "-a", out + "e.4.lcp", "-o", out + "xc", "-l", "250", "-s", "0", "-m", "5"], env=capenv)
    assert r.returncode == 0 and np.fromfile(out + "xc.fq", np.uint8).tobytes() == want, r.stdout[-600:]
    # outputs that cannot be mapped go through pwrite (BFQ_NO_OUTMAP forces that route); no helper threads at all
    for env in (dict(BFQ_NO_OUTMAP="1"), dict(BFQ_PREFAULT_THREADS="0")):
        r = _run([tools["gsufsort"], fq, "--bwt", "--qs", "-o", out + "n"], env=dict(os.environ, **env))
        assert r.returncode == 0 and np.array_equal(np.fromfile(out + "n.bwt", np.uint8), bwt)
        assert np.array_equal(np.fromfile(out + "n.bwt.qs", np.uint8), qs)


def _lease_lines(blob):
    import re
    return [(m.group(1).decode(), float(m.group(2))) for m in re.finditer(rb"\[bfq lease\].*lease (\S+) waited ([0-9.]+) s", blob)]


def test_tools_spread_over_gpus_by_lease(tools, tmp_path, engine):
    """BFQzip_parallel.py:277-285 runs its BFQzip.py children concurrently; their gsufsort / bfq_int processes must not all sit
    on GPU 0.  On this one-GPU box: with BFQ_FAKE_DEVICES=2 two concurrent tools hold two different leases (both map to GPU
    0 here, to two GPUs on a node); with one device they run one after the other.  Outputs stay right either way."""
    fq = str(tmp_path / "in.fastq")
    text = _synth_file(engine, fq, 200_000, 100, seed=9)
    bwt, _, _ = engine.fastq_build_ebwt(text)
    for fake, label in ((2, "spread"), (1, "serial")):
        env = dict(os.environ, BFQ_TRACE="1", BFQ_FAKE_DEVICES=str(fake), BFQ_LEASE_DIR=str(tmp_path))
        ps = [subprocess.Popen([tools["gsufsort"], fq, "--bwt", "--qs", "-o", str(tmp_path / f"{label}{i}")], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for i in range(2)]
        outs = [p.communicate(timeout=300)[0] for p in ps]
        assert all(p.returncode == 0 for p in ps), outs
        leases = [_lease_lines(o) for o in outs]
        assert all(len(x) == 1 for x in leases), outs
        (p0, w0), (p1, w1) = leases[0][0], leases[1][0]
        if fake == 2:
            assert p0 != p1 and max(w0, w1) < 0.2, leases
        else:
            assert p0 == p1 and max(w0, w1) > 0.1, leases      # the loser waited for the winner's whole run
        for i in range(2):
            assert np.array_equal(np.fromfile(str(tmp_path / f"{label}{i}.bwt"), np.uint8), bwt)
    # BFQ_DEVICE pins, BFQ_LEASE=0 switches the lease off
    r = _run([tools["gsufsort"], fq, "--bwt", "--qs", "-o", str(tmp_path / "pin")], env=dict(os.environ, BFQ_TRACE="1", BFQ_DEVICE="0", BFQ_LEASE_DIR=str(tmp_path)))
    assert r.returncode == 0 and len(_lease_lines(r.stdout)) == 1
    r = _run([tools["gsufsort"], fq, "--bwt", "--qs", "-o", str(tmp_path / "nol")], env=dict(os.environ, BFQ_TRACE="1", BFQ_LEASE="0"))
    assert r.returncode == 0 and b"(no lease)" in r.stdout
    r = _run([tools["gsufsort"], fq, "--bwt", "--qs", "-o", str(tmp_path / "bad")], env=dict(os.environ, BFQ_DEVICE="7"))
    assert r.returncode == 1


def test_tools_report_a_full_disk(tools, tmp_path, engine):
    """Outputs that cannot be written (here: a file size limit; a full /dev/shm behaves the same) end in exit status 1 and a
    message -- not in a SIGBUS on the output mapping, and not in a truncated file with status 0 (BFQzip.py:328-336 only
    looks at the status)."""
    import resource, signal
    fq = str(tmp_path / "in.fastq")
    _synth_file(engine, fq, 300_000, 100, seed=3)

    def limit():
        signal.signal(signal.SIGXFSZ, signal.SIG_IGN)                # the write then fails with EFBIG instead of killing the tool
        resource.setrlimit(resource.RLIMIT_FSIZE, (1 << 20, 1 << 20))
    r = subprocess.run([tools["gsufsort"], fq, "--bwt", "--qs", "-o", str(tmp_path / "L")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=300, preexec_fn=limit)
    assert r.returncode == 1 and b"write" in r.stdout.lower(), r.stdout
    assert _run([tools["gsufsort"], fq, "--bwt", "--qs", "-o", str(tmp_path / "OK")]).returncode == 0
    r = subprocess.run([tools["bfq_int"], "-e", str(tmp_path / "OK.bwt"), "-q", str(tmp_path / "OK.bwt.qs"), "-o", str(tmp_path / "L.fq"), "-m", "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, preexec_fn=limit)
    assert r.returncode == 1, r.stdout
