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
    "constant": (1081, "071156a8d39a1e49b4cb657b64cb6130"),
    "dna_like": (99868, "198f37e9978e6d7022925a732193b19f"),
    "empty": (295, "46a75f8b5c04680ceb3ad1b113476a43"),
    "headers": (21406, "734fe491da2dbfe5da76ab7bb966a87b"),
    "one_byte": (305, "0084c84f826c206a864b2623d57518e6"),
    "period4": (1436, "63c7772768b377cc81464f6d3f89fbf1"),
    "random_bytes": (51854, "48d9b8bdaa44addc416ef1c96d3cb79c"),
    "runs_20_symbols": (69617, "ac0c5884c8126f7811057b8cf7110d03"),
    "seg_exact": (9334, "757f8959a65f4e6a59e9086f1d35d7fc"),
    "seg_minus_1": (9334, "e712e6711bb4a08ec3afcbcdbe58f476"),
    "seg_plus_1": (9334, "59d76c1727a07e497dc6abb1e7375533"),
    "smoothed_qs_like": (165361, "53ce8c0eb36a822546e4b97397d22b8a"),
    "two_symbols": (10596, "16a0be7c5a640c0f4581412088dadac3"),
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
    assert len(orc.codec_encode(c["empty"])) == 295
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
    as BFQLINE1; everything else stays BFQRANS1."""
    c = cases()
    names = np.frombuffer(b"".join(b"@A00123:45:HXXXX:1:1101:%d:%d 1:N:0:ACGT\n" % (1000 + i // 7, 2000 + (i * 37) % 9000)
                                   for i in range(50000)), np.uint8)
    for data, kind in ((c["headers"], b"BFQLINE1"), (names, b"BFQLINE1"), (c["dna_like"], b"BFQRANS1"),
                       (np.frombuffer(b"ab\n" * 1000, np.uint8), b"BFQRANS1"),                   # lines too short
                       (np.frombuffer(b"@r1\n@r2", np.uint8), b"BFQRANS1"),                      # no final newline
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
