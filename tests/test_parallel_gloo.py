"""CPU tests of the multi-GPU sharding path: BFQzip_parallel's split rule, the paired
merge rule and the ordered gather, run as 2 processes over gloo.  The per-block
engine is the CPU oracle here (no GPU in this tier); on GPUs the same code path is
driven with bfqzip_amd.api.Engine.run_reads (tests/test_gpu_parity.py)."""
import hashlib, os, sys
import numpy as np
import pytest
from bfqzip_amd import fastq, parallel
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_rule_matches_reference_driver():
    # 100 reads, -t 8 -> 8 blocks of 12, the last one takes 16
    assert parallel.split_blocks(100, 8) == [(12 * i, 12 * i + 12) for i in range(7)] + [(84, 100)]
    assert parallel.split_blocks(100, 2) == [(0, 50), (50, 100)]
    # num_blocks may exceed t: 10 reads, t=4 -> size_block 2 -> 5 blocks
    assert parallel.split_blocks(10, 4) == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10)]
    assert parallel.split_blocks(7, 1) == [(0, 7)]
    assert parallel.split_blocks(0, 3) == []


def _oracle_block(orc):
    p = orc.params(m=5)
    def run(b, q, r):
        ob, oq, st = orc.run_reads(b, q, r, p)
        return ob, oq
    return run


@pytest.mark.parametrize("t,md5", [(2, "d2aac3c45dda67ec3f769273ea6a5568"), (8, "4ada980195fd8d6fb206408c4f892bc6")])
def test_sharded_output_equals_reference_parallel_run(orc, t, md5):
    """md5s: BFQzip_parallel.py example/reads.fastq -t {2,8} -0 driving the compiled reference."""
    b, q, r, h, *_ = util.golden_set("example")
    ob, oq = parallel.run_blocks(_oracle_block(orc), b, q, r, t)
    assert hashlib.md5(fastq.format_fastq(ob, oq, r)).hexdigest() == md5


def test_paired_blocks_rule(orc):
    """-p -t 2: mate block k appended to block k; outputs split back by the block-1 read count."""
    b, q, r, h, *_ = util.golden_set("paired")     # reads_1 followed by reads_2 (100 + 100 reads)
    c1 = parallel.slice_reads(b, q, r, 0, 100); c2 = parallel.slice_reads(b, q, r, 100, 200)
    run = _oracle_block(orc)
    o1b, o1q, o1r, o2b, o2q, o2r = [], [], [0], [], [], [0]
    for bb, bq, br, n1 in parallel.paired_blocks(c1, c2, 2):
        ob, oq = run(bb, bq, br)
        cut = int(br[n1])
        o1b.append(ob[:cut]); o1q.append(oq[:cut]); o2b.append(ob[cut:]); o2q.append(oq[cut:])
        o1r += list(o1r[-1] + (br[1:n1 + 1])); o2r += list(o2r[-1] + (br[n1 + 1:] - br[n1]))
    f1 = fastq.format_fastq(np.concatenate(o1b), np.concatenate(o1q), np.array(o1r, np.uint64))
    f2 = fastq.format_fastq(np.concatenate(o2b), np.concatenate(o2q), np.array(o2r, np.uint64))
    assert hashlib.md5(f1).hexdigest() == "0869c40b37c0d1149f7644025b7bffda"
    assert hashlib.md5(f2).hexdigest() == "26b0df769ae25a5663f953c55d00ab83"


def _worker(rank, world, port, t, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, qq, r, h, *_ = util.golden_set("example")
    res = parallel.run_blocks(_oracle_block(orc), b, qq, r, t, dist=dist)
    if rank == 0:
        q.put(hashlib.md5(fastq.format_fastq(res[0], res[1], r)).hexdigest())
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("t,md5", [(2, "d2aac3c45dda67ec3f769273ea6a5568"), (8, "4ada980195fd8d6fb206408c4f892bc6")])
def test_two_ranks_gloo(t, md5):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + t
    ps = [ctx.Process(target=_worker, args=(rk, 2, port, t, q)) for rk in range(2)]
    for p in ps:
        p.start()
    got = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == md5
