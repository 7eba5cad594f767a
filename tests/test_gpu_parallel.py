"""GPU test of the sharded path: BFQzip_parallel's split, one process per rank (gloo here, both
ranks on the single GPU of the test box), the real engine per block, ordered gather on rank 0.
Expected md5s: the reference's own BFQzip_parallel.py runs."""
import hashlib, os, sys
import pytest
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, t, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from bfqzip_amd import api, fastq, parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = api.Engine(0, m=5)
    def run(b, qq, r):
        ob, oq, st = eng.run_reads(b, qq, r)
        return ob, oq
    b, qq, r, h, *_ = util.golden_set("example")
    res = parallel.run_blocks(run, b, qq, r, t, dist=dist)
    if rank == 0:
        q.put(hashlib.md5(fastq.format_fastq(res[0], res[1], r)).hexdigest())
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("t,md5", [(2, "d2aac3c45dda67ec3f769273ea6a5568"), (8, "4ada980195fd8d6fb206408c4f892bc6")])
def test_two_ranks_real_engine(t, md5):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + t
    ps = [ctx.Process(target=_worker, args=(rk, 2, port, t, q)) for rk in range(2)]
    for p in ps:
        p.start()
    got = q.get(timeout=300)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == md5


def test_parallel_cli_single_process(tmp_path):
    """`python -m bfqzip_amd.parallel` without torchrun = all blocks on GPU 0; same files as BFQzip_parallel.py -t n -0."""
    import subprocess
    fq = os.path.join(util.GOLDEN, "example.fastq")
    out = str(tmp_path / "OUT")
    r = subprocess.run([sys.executable, "-m", "bfqzip_amd.parallel", fq, "-o", out, "-t", "8"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert hashlib.md5(open(out + ".fq", "rb").read()).hexdigest() == "4ada980195fd8d6fb206408c4f892bc6"
    # --m3: the streams BFQzip.py cuts with sed -n 2~4p / 4~4p / 1~4p (BFQzip.py:19-21)
    r = subprocess.run([sys.executable, "-m", "bfqzip_amd.parallel", fq, "-o", out, "-t", "8", "--m3"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    lines = open(out + ".fq", "rb").read().split(b"\n")[:-1]
    assert open(out + ".fq.dna", "rb").read() == b"".join(x + b"\n" for x in lines[1::4])
    assert open(out + ".fq.qs", "rb").read() == b"".join(x + b"\n" for x in lines[3::4])
    assert open(out + ".h", "rb").read() == b"".join(x + b"\n" for x in open(fq, "rb").read().split(b"\n")[:-1][0::4])
    # paired: reads_1 / reads_2 of the reference's example = the two halves of the "paired" golden input
    from bfqzip_amd import fastq, parallel
    b, q, rr, h, *_ = util.golden_set("paired")
    c1 = parallel.slice_reads(b, q, rr, 0, 100); c2 = parallel.slice_reads(b, q, rr, 100, 200)
    f1, f2 = str(tmp_path / "r1.fastq"), str(tmp_path / "r2.fastq")
    open(f1, "wb").write(fastq.format_fastq(*c1, h[:100])); open(f2, "wb").write(fastq.format_fastq(*c2, h[100:]))
    r = subprocess.run([sys.executable, "-m", "bfqzip_amd.parallel", f1, f2, "-p", "-o", out, "-t", "2"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert hashlib.md5(open(out + "_1.fq", "rb").read()).hexdigest() == "0869c40b37c0d1149f7644025b7bffda"
    assert hashlib.md5(open(out + "_2.fq", "rb").read()).hexdigest() == "26b0df769ae25a5663f953c55d00ab83"
