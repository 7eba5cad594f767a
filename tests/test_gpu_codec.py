"""GPU stream codec (bfqzip_amd/csrc/k_codec.hip, SURVEY 8(f).4: the entropy coder that takes the place of step 5,
BFQzip.py:253-275) against its CPU statement oracle/bfq_codec_ref.c: the container must be the same bytes, the decoder
must return the input, on the output streams of the pipeline itself and on the shapes of tests/codec_cases.py.
Parity with 7z PPMd / libbsc is unpinned (external tools, not in the reference tree): the container is this project's own."""
import os, subprocess
import numpy as np
import pytest
from bfqzip_amd import api
from oracle import orc
from tests import util
from tests.codec_cases import cases, sampled_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", sorted(cases().keys()))
def test_container_equals_cpu_statement(engine, name):
    data = cases()[name]
    blob = engine.stream_compress(data)
    want = orc.codec_encode(data)
    assert len(blob) == len(want) and (np.asarray(blob) == want).all()
    back = engine.stream_decompress(blob)
    assert len(back) == len(data) and (np.asarray(back) == data).all()
    assert (orc.codec_decode(np.asarray(blob)) == data).all()


def test_pipeline_streams(engine):
    """OUT.fq.dna / OUT.fq.qs / OUT.h of a 200 k x 100 synthetic run (M=2, --m3 with headers): the smoothed quality stream
    must shrink far below the DNA stream, everything must come back."""
    sp = api.synth_spec(200000, 100, seed=7)
    text = np.empty(200000 * 260, np.uint8)
    n = engine.synth_fastq(sp, text)
    engine.set_params(m=5)
    r = engine.fastq_job([text[:n]], keep_headers=True, fastq=False, streams=True, hdr=True)
    sizes = {}
    for key, raw in (("dna", r.dna), ("qs", r.qs), ("hdr", r.hdr)):
        raw = np.asarray(raw)
        blob = engine.stream_compress(raw)
        assert (np.asarray(engine.stream_decompress(blob)) == raw).all()
        want = orc.codec_encode(raw)
        assert len(blob) == len(want) and (np.asarray(blob) == want).all()
        sizes[key] = (len(raw), len(blob))
    assert sizes["qs"][1] < 0.25 * sizes["qs"][0]            # smoothed qualities: well under 2 bits per value
    assert sizes["dna"][1] < 0.30 * sizes["dna"][0]           # bases: about 2 bits each
    assert sizes["hdr"][1] < 0.35 * sizes["hdr"][0]


def test_job_with_compressed_streams(engine):
    """Steps 1-5 in one call (bfq_fastq_job.compress_streams): the containers hold the streams of the plain job."""
    sp = api.synth_spec(50000, 80, Lmax=120, seed=11)
    text = np.empty(50000 * 300, np.uint8)
    n = engine.synth_fastq(sp, text)
    engine.set_params(m=5)
    plain = engine.fastq_job([text[:n]], keep_headers=True, fastq=True, streams=True, hdr=True)
    z = engine.fastq_job([text[:n]], keep_headers=True, fastq=True, streams=True, hdr=True, compress=True)
    assert z.stats == plain.stats and np.array_equal(z.fastq, plain.fastq)
    for a, b in ((z.dna, plain.dna), (z.qs, plain.qs), (z.hdr, plain.hdr)):
        a = np.asarray(a)
        assert bytes(a[:8]) in (b"BFQRANS2", b"BFQLINE1", b"BFQDNAC1") and len(a) < len(b)
        assert np.array_equal(np.asarray(engine.stream_decompress(a)), np.asarray(b))
        assert np.array_equal(orc.codec_encode(np.asarray(b)), a)
    # two parts (paired blocks): the containers cover the whole collection
    half = int(api.text_nth_newline(text[:n], 4 * 25000 - 1)) + 1
    z2 = engine.fastq_job([text[:half], text[half:n]], fastq=False, streams=True, compress=True)
    assert np.array_equal(np.asarray(engine.stream_decompress(np.asarray(z2.qs))), np.asarray(plain.qs))


def test_ebwt_domain_containers(engine):
    """compress=2: the rows of the edited eBWT instead of the reads.  The containers decode (CPU statement) to an eBWT whose
    inversion by the CPU oracle gives the reads of the plain job; the GPU decoder gives the plain job's line streams."""
    for N, L, Lmax, M, B in ((30000, 100, None, 2, 1), (20000, 40, 90, 1, 0), (5000, 1, 3, 3, 0)):
        sp = api.synth_spec(N, L, Lmax=Lmax, seed=5 + N)
        text = np.empty(N * 300, np.uint8)
        n = engine.synth_fastq(sp, text)
        engine.set_params(m=5, M=M, B=B)
        plain = engine.fastq_job([text[:n]], fastq=False, streams=True)
        z = engine.fastq_job([text[:n]], fastq=False, streams=True, compress=2)
        assert z.stats == plain.stats
        bz, qz = np.asarray(z.dna), np.asarray(z.qs)
        assert bz[:8].tobytes() == b"BFQEBWT1" and len(bz) + len(qz) < len(plain.dna)
        dna, qs, nr = engine.ebwt_decode(bz, qz)
        assert nr == N and np.array_equal(dna, np.asarray(plain.dna)) and np.array_equal(qs, np.asarray(plain.qs))
        # independent of the GPU decoder: the CPU codec decodes the three containers; the rows the walk navigates by
        # (symbols with the replaced ones patched back) invert, on the CPU oracle, to the INPUT reads' bases
        sym_len = int(np.frombuffer(bz[32:40].tobytes(), np.uint64)[0])
        rows_s, rows_p, rows_q = orc.codec_decode(bz[40:40 + sym_len]), orc.codec_decode(bz[40 + sym_len:]), orc.codec_decode(qz)
        assert int((rows_p != 0).sum()) == plain.stats["modified"]
        orig = np.where(rows_p != 0, rows_p, rows_s)
        p = orc.params(K=60000, m=5, M=2, B=0)
        ob, oq, roff, _ = orc.smooth_invert(orig, rows_q, np.zeros(len(orig), np.uint32), p)   # LCP 0: no clusters
        hb, hq, hr = api.synth_host(sp)
        assert np.array_equal(ob, hb) and np.array_equal(roff, hr)
        # mode 3: the same rows for the bases, the qualities as the read-order stream of mode 1
        z3 = engine.fastq_job([text[:n]], fastq=False, streams=True, compress=3)
        z1 = engine.fastq_job([text[:n]], fastq=False, streams=True, compress=1)
        b3, q3 = np.asarray(z3.dna), np.asarray(z3.qs)
        assert np.array_equal(q3, np.asarray(z1.qs)) and np.array_equal(b3[40:], bz[40:]) and b3[28] == 1
        dna3, qs3, nr3 = engine.ebwt_decode(b3, q3)
        assert nr3 == N and np.array_equal(dna3, np.asarray(plain.dna)) and np.array_equal(qs3, np.asarray(plain.qs))
    engine.set_params(m=5, M=2, B=0)


def test_large_stream_round_trip(engine):
    """More segments than one launch has lanes, a model table near its largest size (6 symbols, order 6)."""
    rng = np.random.default_rng(5)
    g = rng.choice(np.frombuffer(b"ACGT", np.uint8), 2_000_000)
    starts = rng.integers(0, len(g) - 150, 400000)
    lines = np.empty((len(starts), 151), np.uint8)
    lines[:, :150] = g[starts[:, None] + np.arange(150)[None, :]]
    lines[:, 150] = 10
    raw = lines.reshape(-1)
    raw[rng.integers(0, len(raw), 5000)] = ord("N")           # 6 symbols (the newline may be hit: any bytes compress)
    blob = engine.stream_compress(raw)
    assert (np.asarray(engine.stream_decompress(blob)) == raw).all()
    assert len(blob) < 0.275 * len(raw)                       # 2 bits per base + N, newlines, 8 bytes per 1024-symbol segment


def test_random_streams(engine):
    """Random alphabets (1..256 symbols), lengths around the segment and group boundaries, skews and run structures:
    the GPU container equals the CPU statement and decodes back."""
    rng = np.random.default_rng(31337)
    for it in range(150):
        A = int(rng.choice([1, 2, 3, 5, 6, 8, 9, 16, 17, 42, 64, 200, 256]))
        n = int(rng.choice([0, 1, 7, 15, 16, 17, 1023, 1024, 1025, 4097, 20000, 70001, 250000]))
        syms = rng.permutation(256)[:A].astype(np.uint8)
        p = rng.dirichlet(np.full(A, float(rng.choice([0.05, 0.3, 1.0, 5.0]))))
        data = rng.choice(syms, n, p=p) if n else np.zeros(0, np.uint8)
        if n and rng.random() < 0.4:                               # runs: repeat every symbol a few times
            data = np.repeat(data, rng.integers(1, 6, n))[:n]
        blob = np.asarray(engine.stream_compress(data))
        want = orc.codec_encode(data)
        assert len(blob) == len(want) and (blob == want).all(), (it, A, n)
        assert (np.asarray(engine.stream_decompress(blob)) == data).all(), (it, A, n)


def test_sampled_model(engine):
    """> 8192 segments: the model comes from every second segment; the container still equals the CPU statement."""
    data = sampled_case()
    blob = np.asarray(engine.stream_compress(data))
    want = orc.codec_encode(data)
    assert len(blob) == len(want) and (blob == want).all()
    assert (np.asarray(engine.stream_decompress(blob)) == data).all()


def test_line_delta_transform(engine):
    """Read names: the GPU transform + container equal the CPU statement's; long / short / unterminated lines stay plain."""
    c = cases()
    rng = np.random.default_rng(8)
    names = np.frombuffer(b"".join(b"@A00123:45:HXXXX:1:1101:%d:%d 1:N:0:ACGT\n" % (1000 + i // 7, 2000 + (i * 37) % 9000)
                                   for i in range(300000)), np.uint8)
    ragged = np.frombuffer(b"".join(b"@" + bytes(rng.integers(97, 100, int(rng.integers(1, 60))).astype(np.uint8)) + b"\n"
                                    for _ in range(20000)), np.uint8)
    for data, kind in ((c["headers"], b"BFQLINE1"), (names, b"BFQLINE1"), (ragged, None), (c["dna_like"], b"BFQRANS2"),
                       (np.frombuffer(b"ab\n" * 1000, np.uint8), b"BFQRANS2"), (np.frombuffer(b"@r1\n@r2", np.uint8), b"BFQRANS2"),
                       (np.frombuffer((b"x" * 20 + b"\n") * 600, np.uint8), b"BFQLINE1")):
        blob = np.asarray(engine.stream_compress(data))
        want = orc.codec_encode(data)
        assert len(blob) == len(want) and (blob == want).all()
        if kind:
            assert blob[:8].tobytes() == kind
        assert (np.asarray(engine.stream_decompress(blob)) == data).all()
    bad = np.asarray(engine.stream_compress(names)).copy()
    bad[24] ^= 1                                                   # the line count of the wrapper
    with pytest.raises(api.BfqError):
        engine.stream_decompress(bad)


def test_members_back_to_back(engine):
    """What the sharded driver writes: one container per block, concatenated."""
    c = cases()
    parts = [c["dna_like"], c["headers"], c["smoothed_qs_like"], c["one_byte"]]
    blob = np.concatenate([np.asarray(engine.stream_compress(p)) for p in parts])
    assert (np.asarray(engine.stream_decompress(blob)) == np.concatenate(parts)).all()


def test_refuses_damaged_streams(engine):
    blob = np.asarray(engine.stream_compress(cases()["dna_like"])).copy()
    with pytest.raises(api.BfqError):
        engine.stream_decompress(blob[:len(blob) // 2])
    bad = blob.copy(); bad[3] ^= 0x20
    with pytest.raises(api.BfqError):
        engine.stream_decompress(bad)
    # payload damage that still parses: refused by the checksum of the decoded bytes (never wrong bytes, never a hang)
    rng = np.random.default_rng(3)
    for _ in range(20):
        flip = blob.copy(); flip[int(rng.integers(len(blob) - 4000, len(blob)))] ^= int(rng.integers(1, 256))
        with pytest.raises(api.BfqError):
            engine.stream_decompress(flip)
    flip = blob.copy(); flip[36] ^= 1                            # the checksum field
    with pytest.raises(api.BfqError):
        engine.stream_decompress(flip)
    # whole segments zeroed (a torn write, a sparse hole inside a .dna payload): k_cdc_decode8's refill must end -- a state
    # word of 0 followed by zeros used to refill for ever
    for a, b in ((3000, 500), (9000, 100), (len(blob) // 2, len(blob) // 4)):
        z = blob.copy(); z[len(blob) - a:len(blob) - b] = 0
        with pytest.raises(api.BfqError):
            engine.stream_decompress(z)
    qs = np.asarray(engine.stream_compress(cases()["smoothed_qs_like"])).copy()     # the generic decoder (> 8 symbols)
    z = qs.copy(); z[len(qs) - 6000:len(qs) - 200] = 0
    with pytest.raises(api.BfqError):
        engine.stream_decompress(z)


def test_read_order_dna_container(engine):
    """BFQDNAC1 on the GPU (k_dnac.hip): chosen for the same streams as the CPU statement chooses it for (the byte equality
    is test_container_equals_cpu_statement's and test_pipeline_streams'); members of both kinds back to back; damage to header
    fields, the lengths' container, payload bytes, zeroed segments and a cut are all refused; BFQ_DNA_STATIC=1 keeps the static
    container; a 2 M-read stream (more than one block of 65536 segments... at this size 64 blocks) round trips."""
    c = cases()
    data = c["reads_30x"]
    blob = np.asarray(engine.stream_compress(data)).copy()
    assert blob[:8].tobytes() == b"BFQDNAC1" and 8 * len(blob) < 0.8 * len(data)
    both = np.concatenate([blob, np.asarray(engine.stream_compress(c["headers"])), np.asarray(engine.stream_compress(c["reads_var_len"]))])
    assert (np.asarray(engine.stream_decompress(both)) == np.concatenate([data, c["headers"], c["reads_var_len"]])).all()
    rng = np.random.default_rng(5)
    for pos in (0, 9, 17, 25, 32, 36, 44, 52, 56, 64, 72 + 40, 72 + 400):
        bad = blob.copy(); bad[pos] ^= 1
        with pytest.raises(api.BfqError):
            engine.stream_decompress(bad)
    for _ in range(20):
        bad = blob.copy(); bad[int(rng.integers(len(blob) - 20000, len(blob)))] ^= int(rng.integers(1, 256))
        with pytest.raises(api.BfqError):
            engine.stream_decompress(bad)
    bad = blob.copy(); bad[len(blob) - 9000:len(blob) - 300] = 0
    with pytest.raises(api.BfqError):
        engine.stream_decompress(bad)
    with pytest.raises(api.BfqError):
        engine.stream_decompress(blob[:len(blob) - 100])
    os.environ["BFQ_DNA_STATIC"] = "1"
    try:
        engine.set_params()                                         # (the environment is read when parameters are set)
        st = np.asarray(engine.stream_compress(data))
        assert st[:8].tobytes() == b"BFQRANS2" and len(st) > 2 * len(blob)
        assert (np.asarray(engine.stream_decompress(st)) == data).all()
    finally:
        del os.environ["BFQ_DNA_STATIC"]
        engine.set_params()
    sp = api.synth_spec(2_000_000, 100, seed=3)
    text = np.empty(2_000_000 * 260, np.uint8)
    n = engine.synth_fastq(sp, text)
    engine.set_params(m=5)
    r = engine.fastq_job([text[:n]], fastq=False, streams=True)
    raw = np.asarray(r.dna)
    z = np.asarray(engine.stream_compress(raw))
    assert z[:8].tobytes() == b"BFQDNAC1" and 8 * len(z) < 0.8 * len(raw)
    assert (np.asarray(engine.stream_decompress(z)) == raw).all()
    engine.set_params()


def test_random_read_streams(engine):
    """Random read-shaped streams (genome size, coverage, read length, fixed / variable lengths with empty lines, error and N
    rates, N runs) around and above the 64 KiB threshold: whichever container the CPU statement chooses (BFQDNAC1, or the
    static one when the stream is short / barely covered / has short lines), the GPU produces the same bytes and decodes them."""
    from tests.codec_cases import reads
    rng = np.random.default_rng(4242)
    kinds = {}
    for it in range(36):
        G = int(rng.choice([800, 3000, 20000, 100000, 600000]))
        L = int(rng.choice([20, 36, 76, 100, 151, 250]))
        cov = int(rng.choice([1, 3, 10, 30, 60]))
        while G * cov > 2_500_000:
            cov = max(1, cov // 2)
        data = reads(rng, G, cov, L, var=bool(rng.integers(0, 2)), n_runs=bool(rng.integers(0, 2)),
                     err=float(rng.choice([0.0, 0.003, 0.01, 0.05])), n_rate=float(rng.choice([0.0, 0.001, 0.02])))
        if not len(data):
            continue
        blob = np.asarray(engine.stream_compress(data))
        want = orc.codec_encode(data)
        assert len(blob) == len(want) and (blob == want).all(), (it, G, L, cov, len(data))
        assert (np.asarray(engine.stream_decompress(blob)) == data).all(), (it, G, L, cov)
        kinds[blob[:8].tobytes()] = kinds.get(blob[:8].tobytes(), 0) + 1
    assert kinds.get(b"BFQDNAC1", 0) >= 8 and kinds.get(b"BFQRANS2", 0) >= 4, kinds


def test_bsc_front_end(tmp_path):
    """`external/libbsc/bsc e F F.bsc -T` as BFQzip.py:265-275 runs it, and `bsc d` back."""
    exe = os.path.join(ROOT, "dropin", "external", "libbsc", "bsc")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bfqzip_amd", "csrc"), "cli"])
    raw = cases()["runs_20_symbols"].tobytes()
    f = tmp_path / "OUT.fq.qs"
    f.write_bytes(raw)
    r = subprocess.run([exe, "e", str(f), str(f) + ".bsc", "-T"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout
    blob = (tmp_path / "OUT.fq.qs.bsc").read_bytes()
    assert blob[:8] == b"BFQRANS2" and len(blob) < len(raw) // 3
    assert (orc.codec_encode(np.frombuffer(raw, np.uint8)) == np.frombuffer(blob, np.uint8)).all()
    r = subprocess.run([exe, "d", str(f) + ".bsc", str(tmp_path / "back")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout
    assert (tmp_path / "back").read_bytes() == raw
    r = subprocess.run([exe, "d", str(f), str(tmp_path / "x")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode != 0
    # eBWT-domain containers back to the line streams (`bsc x`)
    eng = api.Engine(m=5)
    sp = api.synth_spec(3000, 60, seed=3)
    text = np.empty(3000 * 200, np.uint8)
    n = eng.synth_fastq(sp, text)
    plain = eng.fastq_job([text[:n]], fastq=False, streams=True)
    z = eng.fastq_job([text[:n]], fastq=False, streams=True, compress=3)
    (tmp_path / "rows.z").write_bytes(np.asarray(z.dna).tobytes()); (tmp_path / "qs.z").write_bytes(np.asarray(z.qs).tobytes())
    eng.close()
    r = subprocess.run([exe, "x", str(tmp_path / "rows.z"), str(tmp_path / "qs.z"), str(tmp_path / "o.dna"), str(tmp_path / "o.qs")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout
    assert (tmp_path / "o.dna").read_bytes() == np.asarray(plain.dna).tobytes() and (tmp_path / "o.qs").read_bytes() == np.asarray(plain.qs).tobytes()
