"""The CPU statement of the stream codec (oracle/bfq_codec_ref.c): round trips, and the container of fixed
inputs pinned by md5 so that a change of the format cannot go unnoticed (the GPU codec is compared with this statement
byte for byte in tests/test_gpu_codec.py).  Parity with the reference's step-5 tools (7z PPMd, libbsc) is UNPINNED:
neither is part of the reference tree."""
import hashlib
import numpy as np
import pytest
from oracle import orc
from tests.codec_cases import cases

PINNED = {
    "constant": (399, "65a8c3acada189cbf22073e858fc981d"),
    "dna_like": (97858, "7270a674861968d715a74d3c4a1ee4f5"),
    "empty": (293, "b8740b37793b7bf28c0d253e72a37bdb"),
    "headers": (70037, "20a1cdc89cf2b6b75df4a2adfceab305"),
    "one_byte": (303, "6913aa427fbb4706e584abb0fd1c37d9"),
    "period4": (604, "f34093242d5adde9509282de1af54841"),
    "random_bytes": (51030, "d38a69d2fdd49113350261dc61f6dd5b"),
    "runs_20_symbols": (66732, "db9acde1236317cc7ddce3bb41eb851a"),
    "seg_exact": (9197, "92ca46d4a126ed3d802e5057e6bbc99b"),
    "seg_minus_1": (9198, "c4691afee2db6911875a1cc83647fc4a"),
    "seg_plus_1": (9197, "8240501d3f2a8e007e48480d1e54afb8"),
    "smoothed_qs_like": (158634, "5dc9edc7fff082966f48529557766e54"),
    "two_symbols": (10143, "f36f4b03085b86b236142721fdff3050"),
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
    assert len(orc.codec_encode(c["empty"])) == 293
    a = orc.codec_encode(c["dna_like"]); b = orc.codec_encode(c["dna_like"].copy())
    assert (a == b).all()
    # skewed quality-like data: close to its order-0 entropy
    q = c["smoothed_qs_like"]
    p = np.bincount(q, minlength=256) / len(q)
    h = -(p[p > 0] * np.log2(p[p > 0])).sum() * len(q) / 8
    assert len(orc.codec_encode(q)) < 1.2 * h + 20000


def test_damaged_streams_are_refused():
    blob = orc.codec_encode(cases()["dna_like"])
    with pytest.raises(RuntimeError):
        orc.codec_decode(blob[:len(blob) // 2])
    bad = blob.copy(); bad[0] ^= 1
    with pytest.raises(RuntimeError):
        orc.codec_decode(bad)
