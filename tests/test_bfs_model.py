"""CPU test of the algorithm behind k_bfs.hip (LCP array from the eBWT alone): the executable model in tests/bfs_model.py
against LCPs computed from the decoded suffixes."""
from tests import bfs_model


def test_interval_refinement_model():
    tested, bad = bfs_model.check(seed=20240807, iters=400)
    assert tested >= 700 and bad == 0
