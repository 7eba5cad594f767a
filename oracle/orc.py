"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product package (bfqzip_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Params(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("K", "m", "v", "f", "t", "term", "M", "B", "ext")]


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "num_clust", "num_clust_discarded", "num_clust_amb_discarded", "num_clust_mod",
        "num_clust_alleq", "bases_inside", "qs_smoothed", "modified")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_smooth_invert.restype = C.c_int64
        _LIB.orc_invert.restype = C.c_int64
    return _LIB


def params(K=16, m=2, v=ord(">"), f=40, t=20, term=ord("#"), M=2, B=0, ext=0):
    return Params(K, m, v, f, t, term, M, B, ext)


def _p(a, ty=C.c_uint8):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def build_ebwt(bases, quals, roff, term=ord("#"), want_sa=False):
    N = len(roff) - 1
    n = int(roff[-1]) + N
    bases = np.ascontiguousarray(bases, np.uint8); quals = np.ascontiguousarray(quals, np.uint8)
    roff = np.ascontiguousarray(roff, np.uint64)
    bwt = np.empty(n, np.uint8); qs = np.empty(n, np.uint8); lcp = np.empty(n, np.uint32)
    sa = np.empty(n, np.uint64) if want_sa else None
    rc = lib().orc_build_ebwt(_p(bases), _p(quals), _p(roff, C.c_uint64), C.c_uint64(N), term,
                              _p(bwt), _p(qs), _p(lcp, C.c_uint32), _p(sa, C.c_uint64))
    if rc:
        raise RuntimeError(f"orc_build_ebwt rc={rc}")
    return (bwt, qs, lcp, sa) if want_sa else (bwt, qs, lcp)


def smooth_invert(bwt, qs, lcp, p):
    """lcp may be None (bfq_int mode: LCP deduced from the BWT)."""
    n = len(bwt)
    bwt = np.ascontiguousarray(bwt, np.uint8); qs = np.ascontiguousarray(qs, np.uint8)
    if lcp is not None:
        lcp = np.ascontiguousarray(lcp, np.uint32)
    ob = np.empty(n + 1, np.uint8); oq = np.empty(n + 1, np.uint8)
    oroff = np.empty(n + 2, np.uint64)
    st = Stats()
    N = lib().orc_smooth_invert(_p(bwt), _p(qs), _p(lcp, C.c_uint32), C.c_uint64(n), C.byref(p),
                                _p(ob), _p(oq), _p(oroff, C.c_uint64), C.byref(st))
    if N < 0:
        raise RuntimeError(f"orc_smooth_invert rc={N}")
    oroff = oroff[:N + 1].copy()
    tot = int(oroff[-1])
    return ob[:tot].copy(), oq[:tot].copy(), oroff, st.as_dict()


def run_reads(bases, quals, roff, p):
    N = len(roff) - 1
    bases = np.ascontiguousarray(bases, np.uint8); quals = np.ascontiguousarray(quals, np.uint8)
    roff = np.ascontiguousarray(roff, np.uint64)
    ob = np.empty(len(bases) + 1, np.uint8); oq = np.empty(len(bases) + 1, np.uint8)
    st = Stats()
    rc = lib().orc_run_reads(_p(bases), _p(quals), _p(roff, C.c_uint64), C.c_uint64(N), C.byref(p),
                             _p(ob), _p(oq), C.byref(st))
    if rc:
        raise RuntimeError(f"orc_run_reads rc={rc}")
    return ob[:len(bases)].copy(), oq[:len(bases)].copy(), st.as_dict()


def ref_binary(M, B):
    """Path of the compiled REFERENCE bfq_int for (M,B), or None if not built."""
    p = os.path.join(_HERE, "_ref", f"bfq_int_M{M}_B{B}")
    return p if os.path.exists(p) else None


# ---- stream codec (oracle/bfq_codec_ref.c): the CPU statement of the BFQRANS2 container
def codec_encode(data):
    data = np.ascontiguousarray(np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data, np.uint8)
    L = lib()
    L.orc_codec_encode.restype = C.c_int64
    cap = 2 * len(data) + (12 << 20)
    out = np.empty(cap, np.uint8)
    r = L.orc_codec_encode(_p(data), C.c_uint64(len(data)), _p(out), C.c_uint64(cap))
    if r < 0:
        raise RuntimeError("orc_codec_encode failed: %d" % r)
    return out[:r].copy()


def codec_decode(blob):
    """All members of a buffer (one container, or several back to back as the sharded driver writes them)."""
    blob = np.ascontiguousarray(blob, np.uint8)
    L = lib()
    L.orc_codec_decode.restype = C.c_int64
    L.orc_codec_raw_len.restype = C.c_int64
    L.orc_codec_member_len.restype = C.c_int64
    outs = []
    pos = 0
    while True:
        part = blob[pos:]
        ml = L.orc_codec_member_len(_p(part), C.c_uint64(len(part)))
        n = L.orc_codec_raw_len(_p(part), C.c_uint64(len(part)))
        if ml < 0 or n < 0:
            raise RuntimeError("not a BFQRANS2 stream")
        out = np.empty(max(int(n), 1), np.uint8)
        r = L.orc_codec_decode(_p(part), C.c_uint64(ml), _p(out), C.c_uint64(n))
        if r != n:
            raise RuntimeError("orc_codec_decode failed: %d" % r)
        outs.append(out[:n])
        pos += int(ml)
        if pos >= len(blob):
            break
    return outs[0] if len(outs) == 1 else np.concatenate(outs)
