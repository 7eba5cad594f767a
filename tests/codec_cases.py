"""Inputs shared by the codec tests (CPU statement and GPU codec): name -> uint8 array."""
import numpy as np


def cases():
    rng = np.random.default_rng(20261004)
    acgtn = np.frombuffer(b"ACGTN\n", np.uint8)
    out = {
        "empty": np.zeros(0, np.uint8),
        "one_byte": np.array([65], np.uint8),
        "constant": np.full(100000, 66, np.uint8),
        "random_bytes": rng.integers(0, 256, 50000).astype(np.uint8),            # 256 symbols: order 0, binary search in the decoder
        "period4": np.frombuffer(b"ACGT" * 30000, np.uint8).copy(),
        "dna_like": rng.choice(acgtn, 300000, p=[.25, .25, .25, .2, .04, .01]),
        "smoothed_qs_like": rng.choice(np.frombuffer(b"#5?I\n", np.uint8), 1000000, p=[.05, .05, .1, .79, .01]),
        "seg_minus_1": rng.integers(33, 75, 8191).astype(np.uint8),
        "seg_exact": rng.integers(33, 75, 8192).astype(np.uint8),
        "seg_plus_1": rng.integers(33, 75, 8193).astype(np.uint8),
        "two_symbols": rng.integers(0, 2, 70000).astype(np.uint8) * 7 + 40,
    }
    # a quality-like stream with memory (runs), 20 symbols
    q = np.empty(400000, np.uint8)
    cur = 30
    steps = rng.integers(0, 100, len(q))
    jumps = rng.integers(0, 20, len(q))
    for i in range(len(q)):
        if steps[i] < 12:
            cur = int(jumps[i])
        q[i] = 40 + cur
    out["runs_20_symbols"] = q
    # headers
    out["headers"] = np.frombuffer(b"".join(b"@SYN.%d\n" % i for i in range(30000)), np.uint8).copy()
    return out


def sampled_case():
    """150 M symbols: the model is counted on every 8th segment only (S = 8); a symbol ('Z') and a context (a run of 'T')
    that occur ONLY in unsampled segments must still be coded."""
    rng = np.random.default_rng(99)
    n = 150_000_000 + 1234
    a = rng.choice(np.frombuffer(b"ACGT", np.uint8), n, p=[.4, .3, .2, .1])
    seg = 2048                                                     # 150 M / 65536 -> segments of 2048
    a[5 * seg + 100] = ord("Z")
    a[7 * seg + 1000:7 * seg + 1400] = ord("T")
    a[40001 * seg + 7:40001 * seg + 9] = ord("Z")
    return a
