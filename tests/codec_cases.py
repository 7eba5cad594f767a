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
    # read-order DNA (lines of ACGTN): the BFQDNAC1 container when the stream is 64 KiB or more, lines of 15 bases or more
    # on average, none beyond 65535 -- else BFQRANS2
    out["reads_30x"] = reads(rng, 20000, 30, 100)
    out["reads_var_len"] = reads(rng, 6000, 40, 120, var=True)                   # lengths 0..120, empty lines among them
    out["reads_below_64k"] = reads(rng, 2000, 30, 100)[:101 * 640]
    out["reads_short_lines"] = reads(rng, 6000, 40, 12, var=True)                # 6 bases per line on average: not this container
    out["reads_n_runs"] = reads(rng, 4000, 25, 150, n_runs=True)
    out["reads_low_coverage"] = reads(rng, 4000000, 1, 100)                      # nothing to learn: the static container is smaller and is kept
    long_line = rng.choice(np.frombuffer(b"ACGT", np.uint8), 65535)
    out["reads_line_65535"] = np.concatenate([reads(rng, 3000, 40, 100), long_line, np.array([10], np.uint8), reads(rng, 3000, 40, 100)])
    out["reads_line_65536"] = np.concatenate([reads(rng, 3000, 10, 100), long_line, np.array([65, 10], np.uint8)])
    # headers
    out["headers"] = np.frombuffer(b"".join(b"@SYN.%d\n" % i for i in range(30000)), np.uint8).copy()
    return out


def reads(rng, G, cov, L, var=False, n_runs=False, err=0.01, n_rate=0.001):
    """cov-fold coverage of a random genome of G bases by reads of L bases from both strands, one per line."""
    g = rng.integers(0, 4, G).astype(np.uint8)
    comp = np.array([3, 2, 1, 0], np.uint8)
    letters = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for _ in range(G * cov // max(L, 1)):
        l = int(rng.integers(0, L + 1)) if var else L
        p = int(rng.integers(0, G - L))
        s = g[p:p + l].copy()
        if rng.integers(0, 2):
            s = comp[s][::-1].copy()
        e = rng.random(l) < err
        s[e] = (s[e] + rng.integers(1, 4, int(e.sum()))) % 4
        s = letters[s].copy()
        s[rng.random(l) < n_rate] = ord("N")
        if n_runs and rng.integers(0, 10) == 0 and l > 40:
            a = int(rng.integers(0, l - 30))
            s[a:a + int(rng.integers(1, 30))] = ord("N")
        out.append(s.tobytes() + b"\n")
    return np.frombuffer(b"".join(out), np.uint8).copy()


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
