"""CPU tests of the multi-GPU sharding path (bfqzip_amd/parallel.py): BFQzip_parallel's split rule on byte ranges of
memory-mapped files, the paired rule, the per-round size exchange and the offset writes, single process and as 2
processes over gloo.  The per-block engine is the CPU oracle here (tests/util.OracleEngine; no GPU in this tier);
tests/test_gpu_parallel.py drives the same code with bfqzip_amd.api.Engine.
Expected md5s: the reference's own BFQzip_parallel.py runs on its example files."""
import hashlib, os, sys
import numpy as np
import pytest
from bfqzip_amd import fastq, parallel
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLE = os.path.join(util.GOLDEN, "example.fastq")
MD5 = {2: "d2aac3c45dda67ec3f769273ea6a5568", 8: "4ada980195fd8d6fb206408c4f892bc6"}
MD5_P1, MD5_P2 = "0869c40b37c0d1149f7644025b7bffda", "26b0df769ae25a5663f953c55d00ab83"


def md5file(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def paired_inputs(tmp):
    """reads_1 / reads_2 of the reference's example = the two halves of the "paired" golden input."""
    b, q, r, h, *_ = util.golden_set("paired")
    f1, f2 = os.path.join(tmp, "r1.fastq"), os.path.join(tmp, "r2.fastq")
    cut = int(r[100])
    open(f1, "wb").write(fastq.format_fastq(b[:cut], q[:cut], r[:101], h[:100]))
    open(f2, "wb").write(fastq.format_fastq(b[cut:], q[cut:], r[100:] - r[100], h[100:]))
    return f1, f2


def test_split_rule_matches_reference_driver():
    # 100 reads, -t 8 -> 8 blocks of 12, the last one takes 16
    assert parallel.split_blocks(100, 8) == [(12 * i, 12 * i + 12) for i in range(7)] + [(84, 100)]
    assert parallel.split_blocks(100, 2) == [(0, 50), (50, 100)]
    # num_blocks may exceed t: 10 reads, t=4 -> size_block 2 -> 5 blocks
    assert parallel.split_blocks(10, 4) == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10)]
    assert parallel.split_blocks(7, 1) == [(0, 7)]
    assert parallel.split_blocks(7, 0) == [(0, 7)]                 # -t 0: one block (BFQzip_parallel.py:302-303)
    assert parallel.split_blocks(0, 3) == []


def test_line_index_and_byte_blocks():
    from bfqzip_amd import api
    buf = np.frombuffer(open(EXAMPLE, "rb").read(), np.uint8)
    text = buf.tobytes()
    lines = text.split(b"\n")[:-1]
    for chunk in (97, 4096, 1 << 20):
        idx = parallel.TextIndex(buf, parallel.Comm(), api.text_line_counts, api.text_nth_newline, chunk=chunk)
        assert idx.num_lines == 400
        for k in (0, 1, 4, 48, 399, 400):
            assert idx.line_start(k) == sum(len(x) + 1 for x in lines[:k])
        bl = parallel.byte_blocks(idx, 8)
        assert [n for _, _, n in bl] == [12] * 7 + [16] and bl[0][0] == 0 and bl[-1][1] == len(text)
        assert all(bl[i][1] == bl[i + 1][0] for i in range(7))
    # a last line without newline still counts
    idx = parallel.TextIndex(buf[:-1], parallel.Comm(), api.text_line_counts, api.text_nth_newline, chunk=1000)
    assert idx.num_lines == 400 and idx.line_start(400) == len(text) - 1
    assert api.text_nth_newline(b"ab\ncd\n", 1) == 5 and api.text_nth_newline(b"ab", 0) == -1


def test_output_names_follow_the_reference():
    n = parallel.output_names(["d/in.fastq"], "OUT", False)
    assert n[0]["fastq"] == "OUT.fastq" and n[0]["dna"] == "OUT.fastq.dna" and n[0]["hdr"] == "OUT.h"
    n = parallel.output_names(["a.fq", "b.fq"], "OUT", True)
    assert [x["fastq"] for x in n] == ["OUT_1.fq", "OUT_2.fq"]
    n = parallel.output_names(["a.fastq", "b.fastq"], "", True)
    assert [x["fastq"] for x in n] == ["a.cat.fastq", "b.cat.fastq"]          # BFQzip_parallel.py:142-147


def test_sharded_run_with_step_5(orc, tmp_path):
    """--compress: every block's share of every output as one BFQRANS2 container, the files `<name>.bsc` hold them in block
    order and decode to what the run without step 5 writes (paired: the mates' shares are coded separately)."""
    eng = util.OracleEngine(orc, m=5)
    f1, f2 = paired_inputs(str(tmp_path))
    plain = parallel.output_names([f1, f2], str(tmp_path / "P"), True)
    parallel.run_files(eng, parallel.Comm(), [f1, f2], 3, plain, paired=True, headers=True, want_streams=True, want_hdr=True)
    z = parallel.output_names([f1, f2], str(tmp_path / "Z"), True)
    parallel.run_files(eng, parallel.Comm(), [f1, f2], 3, z, paired=True, headers=True, want_streams=True, want_hdr=True, compress=True)
    for o in range(2):
        for kind in ("fastq", "dna", "qs", "hdr"):
            blob = np.frombuffer(open(z[o][kind] + ".bsc", "rb").read(), np.uint8)
            assert blob[:8].tobytes() == b"BFQRANS2"
            assert orc.codec_decode(blob).tobytes() == open(plain[o][kind], "rb").read(), (o, kind)
            assert not os.path.exists(z[o][kind])


@pytest.mark.parametrize("t", [2, 8])
def test_sharded_output_equals_reference_parallel_run(orc, t, tmp_path):
    eng = util.OracleEngine(orc, m=5)
    names = parallel.output_names([EXAMPLE], str(tmp_path / "OUT"), False)
    tot = parallel.run_files(eng, parallel.Comm(), [EXAMPLE], t, names)
    assert md5file(names[0]["fastq"]) == MD5[t]
    assert tot["reads"] == 100 and tot["blocks"] == t


def test_paired_blocks_rule(orc, tmp_path):
    """-p -t 2: mate block k appended to block k; outputs split back by the block-1 read count."""
    f1, f2 = paired_inputs(str(tmp_path))
    eng = util.OracleEngine(orc, m=5)
    names = parallel.output_names([f1, f2], str(tmp_path / "OUT"), True)
    parallel.run_files(eng, parallel.Comm(), [f1, f2], 2, names, paired=True)
    assert md5file(names[0]["fastq"]) == MD5_P1 and md5file(names[1]["fastq"]) == MD5_P2


def test_pile_dealing_is_deterministic_and_balanced():
    cnt = np.zeros((6, 6), np.uint64)
    rng = np.random.default_rng(3)
    cnt[1:, 1:] = rng.integers(1, 1000, (5, 5))
    cnt[4, :] = 0                                                   # no suffix starts with N
    for world in (1, 2, 3, 8):
        a, b = parallel.deal_piles(cnt, world), parallel.deal_piles(cnt.copy(), world)
        assert a == b and sorted(p for r in a for p in r) == sorted((s, s2) for s in (1, 2, 3, 5) for s2 in range(1, 6))
        loads = [sum(int(cnt[s][s2]) for s, s2 in r) for r in a]
        assert max(loads) - min(loads) <= int(cnt.max())             # largest-first to the least loaded rank


def test_global_mode_equals_the_unsharded_run(orc, tmp_path):
    """--global (bfqzip_amd.parallel.run_global, single process): one eBWT over the whole file = the unsharded reference
    run (example/reads.fastq, M2B0 -m 5 -> 29866da0...), and with headers = the -H run (9178301c...)."""
    eng = util.OracleGlobalEngine(orc, m=5)
    names = parallel.output_names([EXAMPLE], str(tmp_path / "G"), False)
    tot = parallel.run_global(eng, parallel.Comm(), [EXAMPLE], names, want_streams=True)
    assert md5file(names[0]["fastq"]) == "29866da058baf8e382927c0023e8ab12"
    assert tot["stats_all_ranks"]["qs_smoothed"] == 4198 and tot["stats_all_ranks"]["modified"] == 10
    eng = util.OracleGlobalEngine(orc, m=5)
    parallel.run_global(eng, parallel.Comm(), [EXAMPLE], names, headers=True, want_streams=True, want_hdr=True)
    assert md5file(names[0]["fastq"]) == "9178301c8ef6c9864d5ebfccf47313d4"
    check_streams(names, [EXAMPLE])


def test_global_mode_paired(orc, tmp_path):
    """Two files = ONE collection (file 1's reads, then file 2's): the result of the unsharded paired run -- which is the
    reference's `-p` with one block -- cut back into OUT_1 / OUT_2; single process and (below) 2 ranks."""
    f1, f2 = paired_inputs(str(tmp_path))
    ref = util.OracleEngine(orc, m=5).fastq_job([open(f1, "rb").read(), open(f2, "rb").read()], keep_headers=True)
    want = (ref.fastq[:ref.part_fastq_off[1]].tobytes(), ref.fastq[ref.part_fastq_off[1]:].tobytes())
    names = parallel.output_names([f1, f2], str(tmp_path / "G"), True)
    parallel.run_global(util.OracleGlobalEngine(orc, m=5), parallel.Comm(), [f1, f2], names, headers=True, want_streams=True, want_hdr=True)
    assert open(names[0]["fastq"], "rb").read() == want[0] and open(names[1]["fastq"], "rb").read() == want[1]
    check_streams(names, [f1, f2])
    assert _run2("global_paired", 2, str(tmp_path)) == hashlib.md5(want[0]).hexdigest() + hashlib.md5(want[1]).hexdigest()


def _streams_of(fq_text, in_text):
    lines = fq_text.split(b"\n")[:-1]
    return (b"".join(x + b"\n" for x in lines[1::4]), b"".join(x + b"\n" for x in lines[3::4]),
            b"".join(x + b"\n" for x in in_text.split(b"\n")[:-1][0::4]))


def check_streams(names, inputs):
    for nm, inp in zip(names, inputs):
        fq = open(nm["fastq"], "rb").read()
        dna, qs, hdr = _streams_of(fq, open(inp, "rb").read())
        assert open(nm["dna"], "rb").read() == dna and open(nm["qs"], "rb").read() == qs
        assert open(nm["hdr"], "rb").read() == hdr
        assert fq.split(b"\n")[:-1][0::4] == hdr.split(b"\n")[:-1]              # --m3 keeps the header lines


def _worker(rank, world, port, mode, t, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = parallel.Comm(dist)
    eng = util.OracleEngine(orc, m=5)
    try:
        if mode == "single":
            names = parallel.output_names([EXAMPLE], os.path.join(tmp, "OUT"), False)
            parallel.run_files(eng, comm, [EXAMPLE], t, names)
            res = md5file(names[0]["fastq"])
        elif mode == "global":
            names = parallel.output_names([EXAMPLE], os.path.join(tmp, "G"), False)
            parallel.run_global(util.OracleGlobalEngine(orc, m=5), comm, [EXAMPLE], names)
            res = md5file(names[0]["fastq"])
        elif mode == "global_paired":
            f1, f2 = os.path.join(tmp, "r1.fastq"), os.path.join(tmp, "r2.fastq")
            names = parallel.output_names([f1, f2], os.path.join(tmp, "G2"), True)
            parallel.run_global(util.OracleGlobalEngine(orc, m=5), comm, [f1, f2], names, headers=True)
            res = md5file(names[0]["fastq"]) + md5file(names[1]["fastq"])
        elif mode == "m3":
            names = parallel.output_names([EXAMPLE], os.path.join(tmp, "OUT"), False)
            parallel.run_files(eng, comm, [EXAMPLE], t, names, headers=True, want_streams=True, want_hdr=True)
            if rank == 0:
                check_streams(names, [EXAMPLE])
            res = "ok"
        else:
            f1, f2 = os.path.join(tmp, "r1.fastq"), os.path.join(tmp, "r2.fastq")
            names = parallel.output_names([f1, f2], os.path.join(tmp, "OUT"), True)
            hdrs = mode == "paired_m3"
            parallel.run_files(eng, comm, [f1, f2], t, names, paired=True, headers=hdrs, want_streams=hdrs, want_hdr=hdrs)
            if hdrs and rank == 0:
                check_streams(names, [f1, f2])
            res = md5file(names[0]["fastq"]) + md5file(names[1]["fastq"])
        if rank == 0:
            q.put(res)
    except Exception as e:                                     # surface the failure instead of a queue timeout
        q.put(f"rank {rank}: {type(e).__name__}: {e}")
        raise
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _run2(mode, t, tmp):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + t + len(mode) * 13) % 3000
    ps = [ctx.Process(target=_worker, args=(rk, 2, port, mode, t, tmp, q)) for rk in range(2)]
    for p in ps:
        p.start()
    got = q.get(timeout=180)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0, got
    return got


@pytest.mark.parametrize("t", [2, 8])
def test_two_ranks_gloo(t, tmp_path):
    assert _run2("single", t, str(tmp_path)) == MD5[t]


def test_two_ranks_gloo_global_mode(tmp_path):
    """2 ranks, one collection: text exchange, pile dealing, delta all-reduce, offset writes -> the unsharded result."""
    assert _run2("global", 2, str(tmp_path)) == "29866da058baf8e382927c0023e8ab12"


def test_two_ranks_gloo_streams_and_headers(tmp_path):
    assert _run2("m3", 8, str(tmp_path)) == "ok"


def test_two_ranks_gloo_paired(tmp_path):
    paired_inputs(str(tmp_path))
    assert _run2("paired", 2, str(tmp_path)) == MD5_P1 + MD5_P2
    # headers + streams in paired mode: bases / qualities are those of the header-less run, header lines kept
    assert _run2("paired_m3", 2, str(tmp_path)) != ""
    for k, md in ((1, MD5_P1), (2, MD5_P2)):
        fq = open(str(tmp_path / f"OUT_{k}.fastq"), "rb").read().split(b"\n")[:-1]
        bare = b"".join((b"@" if i % 4 == 0 else x) + b"\n" for i, x in enumerate(fq))
        assert hashlib.md5(bare).hexdigest() == md
