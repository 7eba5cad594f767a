"""The CPU statement of the stream codec (oracle/bfq_codec_ref.c): round trips, and the container of fixed
inputs pinned by md5 so that a change of the format cannot go unnoticed (the GPU codec is compared with this statement
byte for byte in tests/test_gpu_codec.py).  Parity with the reference's step-5 tools (7z PPMd, libbsc) is UNPINNED:
neither is part of the reference tree."""
import hashlib
import numpy as np
import pytest
from oracle import orc
from tests.codec_cases import cases, sampled_case

PINNED = {
    "constant": (1089, "93ac54242ad00ddffd026b4d461aaf8d"),
    "dna_like": (85710, "8908436e794a063d21d29ff841ba85c5"),
    "empty": (303, "2ff219f3724e3a348c2f26ce6113ec0b"),
    "headers": (21414, "805f6a3ac5b733d6bf3ed344f6631c7e"),
    "one_byte": (313, "2905befabdb22bfbf521f622c0d919b3"),
    "period4": (1294, "cccae62e066d6fa97bb415f3314ad7de"),
    "random_bytes": (51862, "32abd53a601caad662bde30ac9e35ea7"),
    "runs_20_symbols": (54243, "7536cee1630653bfaff7ec6406863958"),
    "reads_30x": (52261, "3d504c44d3c9863a41b4f05c8af63b00"),
    "reads_below_64k": (16433, "b8ea7c3998fc519d3b9c8adadcb83b18"),
    "reads_line_65535": (51383, "9a723ff6f6b4f7123749bb7a7773c153"),
    "reads_line_65536": (25315, "062933d188e2438faade821ed829737d"),
    "reads_low_coverage": (1074049, "fce77501c8f0e2f8cba4d4fc8949629c"),
    "reads_n_runs": (10703, "692839123ec4f9ee9278599e3c7b66f6"),
    "reads_short_lines": (41979, "732781ecf67bbe82ab8bd6699ac107b4"),
    "reads_var_len": (19100, "e66711b21d0392e044e572a830420284"),
    "seg_exact": (6049, "1bd13eb6b4cf059eba4e312708485bbf"),
    "seg_minus_1": (6047, "9cf61b80f288462b74feac5f7383c40d"),
    "seg_plus_1": (6055, "4ee892eeabdab26f610a16cce132ec25"),
    "smoothed_qs_like": (145220, "84d5696f061dd2649555f4469c95fae0"),
    "two_symbols": (9581, "7883b1cea8d1738db45801b133237b43"),
}


@pytest.mark.parametrize("name", sorted(cases().keys()))
def test_round_trip(name):
    data = cases()[name]
    blob = orc.codec_encode(data)
    back = orc.codec_decode(blob)
    assert len(back) == len(data) and (back == data).all()
    if len(data) >= 50000 and name not in ("random_bytes",):
        assert len(blob) < len(data)          # everything but noise shrinks


@pytest.mark.parametrize("name", sorted(PINNED.keys()))
def test_pinned_containers(name):
    blob = orc.codec_encode(cases()[name])
    assert (len(blob), hashlib.md5(blob.tobytes()).hexdigest()) == PINNED[name]


def test_container_is_deterministic_and_sized():
    c = cases()
    assert len(orc.codec_encode(c["empty"])) == 303
    a = orc.codec_encode(c["dna_like"]); b = orc.codec_encode(c["dna_like"].copy())
    assert (a == b).all()
    # skewed quality-like data: close to its order-0 entropy
    q = c["smoothed_qs_like"]
    p = np.bincount(q, minlength=256) / len(q)
    h = -(p[p > 0] * np.log2(p[p > 0])).sum() * len(q) / 8
    assert len(orc.codec_encode(q)) < 1.2 * h + 20000


def test_sampled_model_codes_what_the_sample_missed():
    data = sampled_case()
    blob = orc.codec_encode(data)
    assert (orc.codec_decode(blob) == data).all()
    assert len(blob) < 0.245 * len(data)          # entropy of the source: 1.85 bits per symbol


def test_line_delta_transform_of_read_names():
    """Streams of short lines (8 .. 128 bytes on average) that shrink to 3/4 or less under the line-delta transform travel
    as BFQLINE1; everything else stays BFQRANS2."""
    c = cases()
    names = np.frombuffer(b"".join(b"@A00123:45:HXXXX:1:1101:%d:%d 1:N:0:ACGT\n" % (1000 + i // 7, 2000 + (i * 37) % 9000)
                                   for i in range(50000)), np.uint8)
    for data, kind in ((c["headers"], b"BFQLINE1"), (names, b"BFQLINE1"), (c["dna_like"], b"BFQRANS2"),
                       (np.frombuffer(b"ab\n" * 1000, np.uint8), b"BFQRANS2"),                   # lines too short
                       (np.frombuffer(b"@r1\n@r2", np.uint8), b"BFQRANS2"),                      # no final newline
                       (np.frombuffer((b"x" * 20 + b"\n") * 600, np.uint8), b"BFQLINE1")):       # identical lines, more than one group
        blob = orc.codec_encode(data)
        assert blob[:8].tobytes() == kind
        assert (orc.codec_decode(blob) == data).all()
    assert len(orc.codec_encode(names)) < len(names) // 12


def test_members_back_to_back():
    c = cases()
    parts = [c["dna_like"], c["empty"], c["headers"], c["one_byte"]]
    blob = np.concatenate([orc.codec_encode(p) for p in parts])
    assert (orc.codec_decode(blob) == np.concatenate(parts)).all()


def test_damaged_streams_are_refused():
    blob = orc.codec_encode(cases()["dna_like"])
    with pytest.raises(RuntimeError):
        orc.codec_decode(blob[:len(blob) // 2])
    bad = blob.copy(); bad[0] ^= 1
    with pytest.raises(RuntimeError):
        orc.codec_decode(bad)


def test_checksum_and_damage_are_noticed():
    """The container carries a checksum of the raw bytes (7z and bsc, whose place this takes, verify a CRC): a payload that
    still parses but decodes to other bytes is refused; so is a segment that was zeroed as a whole (a torn write, a sparse
    hole) -- the decoder must come back, not spin on a state that stays 0."""
    data = cases()["dna_like"]
    blob = orc.codec_encode(data).copy()
    assert (orc.codec_decode(blob) == data).all()
    b2 = blob.copy(); b2[36] ^= 1                                   # the checksum field itself
    with pytest.raises(RuntimeError):
        orc.codec_decode(b2)
    flips = 0
    rng = np.random.default_rng(3)
    for _ in range(40):                                             # payload damage: whatever still parses must fail the checksum
        b3 = blob.copy(); b3[int(rng.integers(len(blob) - 4000, len(blob)))] ^= int(rng.integers(1, 256))
        with pytest.raises(RuntimeError):
            orc.codec_decode(b3)
        flips += 1
    assert flips == 40
    b4 = blob.copy(); b4[len(blob) - 3000:len(blob) - 500] = 0      # more than one whole segment of zeros
    with pytest.raises(RuntimeError):
        orc.codec_decode(b4)


def test_read_order_dna_container():
    """BFQDNAC1 (block-adaptive hashed context model): what it is chosen for, what it must refuse."""
    c = cases()
    for name, kind in (("reads_30x", b"BFQDNAC1"), ("reads_var_len", b"BFQDNAC1"), ("reads_n_runs", b"BFQDNAC1"), ("reads_line_65535", b"BFQDNAC1"),
                       ("reads_below_64k", b"BFQRANS2"), ("reads_low_coverage", b"BFQRANS2"), ("reads_short_lines", b"BFQRANS2"), ("reads_line_65536", b"BFQRANS2"), ("dna_like", b"BFQRANS2")):
        assert orc.codec_encode(c[name])[:8].tobytes() == kind, name
    data = c["reads_30x"]
    blob = orc.codec_encode(data).copy()
    assert 8 * len(blob) < 0.8 * len(data)                          # 30x coverage: well under one bit per base (static order-k model: 2.0)
    # members of both kinds back to back
    both = np.concatenate([blob, orc.codec_encode(c["headers"]), orc.codec_encode(c["reads_var_len"])])
    assert (orc.codec_decode(both) == np.concatenate([data, c["headers"], c["reads_var_len"]])).all()
    # damage: header fields, the lengths' container, payload bytes, zeroed segments, a cut
    rng = np.random.default_rng(5)
    for pos in (0, 9, 17, 25, 32, 36, 44, 52, 56, 64, 72 + 40, 72 + 400):
        bad = blob.copy(); bad[pos] ^= 1
        with pytest.raises(RuntimeError):
            orc.codec_decode(bad)
    for _ in range(30):
        bad = blob.copy(); bad[int(rng.integers(len(blob) - 20000, len(blob)))] ^= int(rng.integers(1, 256))
        with pytest.raises(RuntimeError):
            orc.codec_decode(bad)
    bad = blob.copy(); bad[len(blob) - 9000:len(blob) - 300] = 0
    with pytest.raises(RuntimeError):
        orc.codec_decode(bad)
    with pytest.raises(RuntimeError):
        orc.codec_decode(blob[:len(blob) - 100])
