"""GPU tests of the sharded path (bfqzip_amd/parallel.py): BFQzip_parallel's split on byte ranges, one process per
rank (gloo here, both ranks on the single GPU of the test box), the real engine per block (bfq_fastq_run_job),
outputs written at their final offsets.
Expected md5s: the reference's own BFQzip_parallel.py runs; the large paired case (BASELINE
config 4's shape: -p, M=1, --m3 streams, headers) is checked block by block against the oracle."""
import hashlib, os, subprocess, sys
import numpy as np
import pytest
from tests import util
from tests.test_parallel_gloo import MD5, MD5_P1, MD5_P2, EXAMPLE, md5file, paired_inputs, check_streams

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, args, tmp, timeout=900, backend="gloo"):
    """`python -m torch.distributed.run --nproc-per-node world -m bfqzip_amd.parallel ...` over gloo (one GPU);
    backend=None: the driver's own default (nccl = RCCL, one GPU per rank)."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("BFQ_BACKEND", None)
    if backend:
        env["BFQ_BACKEND"] = backend
    port = 29700 + (os.getpid() * 11 + len(" ".join(args))) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "bfqzip_amd.parallel"] + args
    if world == 1:
        cmd = [sys.executable, "-m", "bfqzip_amd.parallel"] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    return r.stdout.decode()


@pytest.mark.parametrize("world,t", [(1, 8), (2, 2), (2, 8)])
def test_sharded_runs_match_reference_driver(world, t, tmp_path):
    out = str(tmp_path / "OUT")
    _launch(world, [EXAMPLE, "-o", out, "-t", str(t), "-0"], str(tmp_path))
    assert md5file(out + ".fastq") == MD5[t]


def test_streams_headers_and_paired(tmp_path):
    out = str(tmp_path / "OUT")
    # --m3: the streams BFQzip.py cuts with sed -n 2~4p / 4~4p / 1~4p (BFQzip.py:19-21), header lines kept
    _launch(2, [EXAMPLE, "-o", out, "-t", "8", "--m3"], str(tmp_path))
    from bfqzip_amd import parallel
    check_streams(parallel.output_names([EXAMPLE], out, False), [EXAMPLE])
    # paired
    f1, f2 = paired_inputs(str(tmp_path))
    _launch(2, [f1, f2, "-p", "-o", out, "-t", "2"], str(tmp_path))
    assert md5file(out + "_1.fastq") == MD5_P1 and md5file(out + "_2.fastq") == MD5_P2
    _launch(1, [f1, f2, "-p", "-o", out, "-t", "2", "--m3", "--pinned"], str(tmp_path))
    check_streams(parallel.output_names([f1, f2], out, True), [f1, f2])
    for k, md in ((1, MD5_P1), (2, MD5_P2)):
        fq = open(out + f"_{k}.fastq", "rb").read().split(b"\n")[:-1]
        assert hashlib.md5(b"".join((b"@" if i % 4 == 0 else x) + b"\n" for i, x in enumerate(fq))).hexdigest() == md


def test_step_5_in_the_sharded_run(orc, tmp_path):
    """--compress over 2 ranks: `<name>.bsc` = one BFQRANS2 container per block in block order; decoded (CPU statement and
    GPU codec) they are the files of the run without step 5."""
    from bfqzip_amd import api
    f1, f2 = paired_inputs(str(tmp_path))
    plain, z = str(tmp_path / "P"), str(tmp_path / "Z")
    _launch(2, [f1, f2, "-p", "-o", plain, "-t", "3", "--m3"], str(tmp_path))
    _launch(2, [f1, f2, "-p", "-o", z, "-t", "3", "--m3", "--compress"], str(tmp_path))
    eng = api.Engine()
    for suffix in ("_1.fastq", "_1.fastq.dna", "_1.fastq.qs", "_1.h", "_2.fastq", "_2.fastq.qs", "_2.h"):
        want = open(plain + suffix, "rb").read()
        blob = np.frombuffer(open(z + suffix + ".bsc", "rb").read(), np.uint8)
        assert orc.codec_decode(blob).tobytes() == want, suffix
        assert np.asarray(eng.stream_decompress(blob)).tobytes() == want, suffix
    eng.close()


def _synth_fastq(path, spec_seed, n_reads, L, tag):
    """Seeded synthetic reads as a FASTQ file with distinct header lines of varying length."""
    from bfqzip_amd import api, fastq
    b, q, r = api.synth_host(api.synth_spec(n_reads, L, seed=spec_seed))
    idx = np.arange(n_reads)
    hdr = np.char.add(np.char.add("@SYN.", idx.astype(str)), np.where(idx % 3 == 0, f" {tag} len={L}", f"/{tag}"))
    h = fastq.HeaderSpans.from_list([x.encode() for x in hdr.tolist()])
    open(path, "wb").write(fastq.format_fastq(b, q, r, h))


def test_paired_m1_streams_3M_reads_two_ranks(orc, tmp_path):
    """BASELINE config 4's shape at test size: paired-end, M=1 (mean-error smoothing), --m3 (DNA / QS / header streams,
    header lines kept), 2 x 1.6 M reads of 40 bp, -t 8 (8 blocks of 200 k + 200 k reads), 2 ranks.
    Blocks 0, 3 and 7 are recomputed by the oracle and compared byte for byte with their share of every output file;
    all blocks: sizes, header lines and stream/FASTQ consistency."""
    from bfqzip_amd import parallel, api, fastq
    N, L, T = 1_600_000, 40, 8
    f1, f2 = str(tmp_path / "s_1.fastq"), str(tmp_path / "s_2.fastq")
    _synth_fastq(f1, 777, N, L, "1"); _synth_fastq(f2, 778, N, L, "2")
    out = str(tmp_path / "OUT")
    # --pinned: page-locked output buffers = the path that copies finished read ranges out while the inversion goes on
    _launch(2, [f1, f2, "-p", "-o", out, "-t", str(T), "--m3", "--M", "1", "--pinned"], str(tmp_path), timeout=1500)
    names = parallel.output_names([f1, f2], out, True)
    check_streams(names, [f1, f2])                                                  # all blocks: streams == sed of the FASTQ
    ins = [np.fromfile(f, np.uint8) for f in (f1, f2)]
    comm = parallel.Comm()
    blocks = [parallel.byte_blocks(parallel.TextIndex(a, comm, api.text_line_counts, api.text_nth_newline), T) for a in ins]
    assert len(blocks[0]) == T and all(n == N // T for _, _, n in blocks[0])
    outs = [{k: np.fromfile(nm[k], np.uint8) for k in ("fastq", "dna", "qs", "hdr")} for nm in names]
    for o in range(2):
        assert len(outs[o]["fastq"]) == len(ins[o]) and len(outs[o]["dna"]) == N * (L + 1)   # records keep their size
    eng = util.OracleEngine(orc, m=5, M=1)
    for k in (0, 3, T - 1):
        parts = [ins[o][blocks[o][k][0]:blocks[o][k][1]] for o in range(2)]
        ref = eng.fastq_job(parts, keep_headers=True, fastq=True, streams=True, hdr=True)
        assert ref.stats["qs_smoothed"] > 0
        for o in range(2):
            b0, b1, _ = blocks[o][k]
            s0 = k * (N // T) * (L + 1)
            h0 = int(np.count_nonzero(ins[o][:b0] == 10)) // 4                    # reads before the block
            hoff = len(b"".join(x + b"\n" for x in bytes(ins[o][:b0]).split(b"\n")[0::4][:h0]))
            lo, hi = ref.part_fastq_off[o], ref.part_fastq_off[o + 1]
            assert np.array_equal(outs[o]["fastq"][b0:b1], ref.fastq[lo:hi]), (k, o, "fastq")
            lo, hi = ref.part_stream_off[o], ref.part_stream_off[o + 1]
            assert np.array_equal(outs[o]["dna"][s0:s0 + hi - lo], ref.dna[lo:hi]), (k, o, "dna")
            assert np.array_equal(outs[o]["qs"][s0:s0 + hi - lo], ref.qs[lo:hi]), (k, o, "qs")
            lo, hi = ref.part_hdr_off[o], ref.part_hdr_off[o + 1]
            assert np.array_equal(outs[o]["hdr"][hoff:hoff + hi - lo], ref.hdr[lo:hi]), (k, o, "hdr")


def test_global_mode_gives_the_unsharded_result(tmp_path):
    """--global: ONE eBWT over the whole input, its two-symbol piles dealt to the ranks (k_global.hip, position-mode clusters),
    edits combined by an all-reduce: the result must be that of the unsharded run -- the reference's own md5 on its example
    (SURVEY App. B), and the single-engine result on 200 k synthetic reads (M=1, B=1, headers, two-frequent-symbol sites)."""
    from bfqzip_amd import api, parallel
    out = str(tmp_path / "G")
    for world in (1, 2):
        _launch(world, [EXAMPLE, "-o", out, "--global", "-v", "1"], str(tmp_path))
        assert md5file(out + ".fastq") == "29866da058baf8e382927c0023e8ab12"
    _launch(2, [EXAMPLE, "-o", out, "--global", "--m3"], str(tmp_path))
    assert md5file(out + ".fastq") == "9178301c8ef6c9864d5ebfccf47313d4"                 # = bfq_int -H
    check_streams(parallel.output_names([EXAMPLE], out, False), [EXAMPLE])
    # fewer reads than ranks (an empty block), an empty file
    one = str(tmp_path / "one.fastq"); open(one, "wb").write(b"@x\nACGTTGCA\n+\nIIIIIIII\n")
    _launch(2, [one, "-o", out, "--global"], str(tmp_path))
    assert open(out + ".fastq", "rb").read() == b"@\nACGTTGCA\n+\nIIIIIIII\n"
    nil = str(tmp_path / "nil.fastq"); open(nil, "wb").close()
    _launch(2, [nil, "-o", out, "--global"], str(tmp_path))
    assert open(out + ".fastq", "rb").read() == b""
    f = str(tmp_path / "syn.fastq")
    b, q, r = api.synth_host(api.synth_spec(200_000, 30, Lmax=70, seed=99, coverage=40, err_ppm=20000, n_ppm=8000, snp_every=97, dsnp_every=131))
    from bfqzip_amd import fastq
    idx = np.arange(len(r) - 1)
    h = fastq.HeaderSpans.from_list([x.encode() for x in np.char.add("@R", idx.astype(str)).tolist()])
    open(f, "wb").write(fastq.format_fastq(b, q, r, h))
    eng = api.Engine(0, m=5, M=1, B=1)
    ref = eng.fastq_job([open(f, "rb").read()], keep_headers=True, fastq=True, streams=True, hdr=True)
    assert ref.stats["num_clust_mod"] > 0 and ref.stats["modified"] > 0
    want = {"fastq": ref.fastq.tobytes(), "dna": ref.dna.tobytes(), "qs": ref.qs.tobytes(), "hdr": ref.hdr.tobytes()}
    eng.close()
    for world in (1, 2):
        _launch(world, [f, "-o", out, "--global", "--m3", "--M", "1", "--B", "1"], str(tmp_path))
        nm = parallel.output_names([f], out, False)[0]
        for k in want:
            assert open(nm[k], "rb").read() == want[k], (world, k)


def test_global_mode_paired(tmp_path):
    """--global -p: two files = ONE collection (file 1's reads, then file 2's); the two outputs are the shares of the unsharded
    paired run = one bfq_fastq_run_job over both files (what the reference's -p with a single block computes)."""
    from bfqzip_amd import api, parallel, fastq
    out = str(tmp_path / "G")
    f1, f2 = paired_inputs(str(tmp_path))
    g1, g2 = str(tmp_path / "s1.fastq"), str(tmp_path / "s2.fastq")
    for path, seed in ((g1, 41), (g2, 42)):
        b, q, r = api.synth_host(api.synth_spec(100_000, 40, Lmax=90, seed=seed, coverage=30, err_ppm=15000, snp_every=97, dsnp_every=131))
        idx = np.arange(len(r) - 1)
        h = fastq.HeaderSpans.from_list([x.encode() for x in np.char.add("@P%d." % seed, idx.astype(str)).tolist()])
        open(path, "wb").write(fastq.format_fastq(b, q, r, h))
    for (a, b), par, extra in (((f1, f2), dict(m=5), []), ((g1, g2), dict(m=5, M=3, B=1), ["--M", "3", "--B", "1"])):
        eng = api.Engine(0, **par)
        ref = eng.fastq_job([open(a, "rb").read(), open(b, "rb").read()], keep_headers=True, fastq=True, streams=True, hdr=True)
        eng.close()
        want = []
        for o in range(2):
            want.append({"fastq": ref.fastq[ref.part_fastq_off[o]:ref.part_fastq_off[o + 1]].tobytes(),
                         "dna": ref.dna[ref.part_stream_off[o]:ref.part_stream_off[o + 1]].tobytes(),
                         "qs": ref.qs[ref.part_stream_off[o]:ref.part_stream_off[o + 1]].tobytes(),
                         "hdr": ref.hdr[ref.part_hdr_off[o]:ref.part_hdr_off[o + 1]].tobytes()})
        for world in (1, 2):
            _launch(world, [a, b, "-p", "-o", out, "--global", "--m3"] + extra, str(tmp_path))
            names = parallel.output_names([a, b], out, True)
            for o in range(2):
                for k in want[o]:
                    assert open(names[o][k], "rb").read() == want[o][k], (world, o, k)


def _gpus():
    try:
        import torch
        return torch.cuda.device_count()                             # counting does not initialise the GPU
    except Exception:
        return 0


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs: the RCCL branches of parallel.py (one rank per GPU)")
def test_two_gpus_over_rccl(tmp_path):
    """The default backend (nccl = RCCL over xGMI), one GPU per rank: the sharded run, the paired run and the global mode
    (broadcasts of the text, all-reduce of the deltas on uint8 device tensors) reproduce the reference md5s.  Skipped on the
    one-GPU test boxes; there the same code runs over gloo (tests above)."""
    out = str(tmp_path / "OUT")
    _launch(2, [EXAMPLE, "-o", out, "-t", "2", "-0"], str(tmp_path), backend=None)
    assert md5file(out + ".fastq") == MD5[2]
    _launch(2, [EXAMPLE, "-o", out, "-t", "8", "-0"], str(tmp_path), backend=None)
    assert md5file(out + ".fastq") == MD5[8]
    f1, f2 = paired_inputs(str(tmp_path))
    _launch(2, [f1, f2, "-p", "-o", out, "-t", "2"], str(tmp_path), backend=None)
    assert md5file(out + "_1.fastq") == MD5_P1 and md5file(out + "_2.fastq") == MD5_P2
    g = str(tmp_path / "G")
    _launch(2, [EXAMPLE, "-o", g, "--global", "-0"], str(tmp_path), backend=None)
    assert md5file(g + ".fastq") == "29866da058baf8e382927c0023e8ab12"                  # the reference's UNSHARDED run
