"""Host-side mirror of the reference's interface for the hot path.

The reference exposes this path only as executables driven by BFQzip.py /
BFQzip_ext.py (gsufsort|eGap -> bfq_int|bfq_ext).  `Engine` offers the same
operations as functions over numpy arrays, all executed by libbfqhip.so on the
GPU (no CPU fallback):

    build_ebwt(...)     ~ gsufsort <fq> --bwt --qs -o OUT        (BFQzip.py:184)
                        ~ eGap <fq> --qs --lcp --lbytes 1 -o OUT  (BFQzip_ext.py:177)
    smooth_invert(...)  ~ bfq_int -e OUT.bwt -q OUT.bwt.qs -o OUT.fq -m 5 ...   (BFQzip.py:215-222)
                        ~ bfq_ext ... -a OUT.1.lcp                (BFQzip_ext.py:208-214)
    run_reads(...)      = both, fused, nothing written in between

Parameter names and defaults are those of bfq_int's getopt flags
(bfq_int.cpp:883-935) plus the compile-time knobs M and B.
"""
import ctypes as C
import numpy as np
from . import _lib


class BfqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbfqhip error {code}: {msg}")
        self.code = code


def make_params(k=16, m=2, v=ord(">"), f=40, t=20, s=ord("#"), M=2, B=0, ext=0, piles=0, ws_cap_mib=0):
    p = _lib.Params()
    p.K, p.m, p.v, p.f, p.t, p.term, p.M, p.B, p.ext, p.piles, p.ws_cap_mib = k, m, v, f, t, s, M, B, ext, piles, ws_cap_mib
    return p


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def _u8(text):
    """bytes / bytearray / ndarray / memmap slice -> contiguous uint8 ndarray (no copy when possible)."""
    if isinstance(text, np.ndarray):
        return text if (text.dtype == np.uint8 and text.flags.c_contiguous) else np.ascontiguousarray(text, np.uint8)
    return np.frombuffer(text, np.uint8)


class PinnedBuffer:
    """Page-locked host memory (bfq_host_alloc) as a uint8 numpy array: transferred by direct DMA."""

    def __init__(self, nbytes):
        self.L = _lib.lib()
        self.nbytes = int(nbytes)
        self.ptr = self.L.bfq_host_alloc(max(self.nbytes, 1))
        if not self.ptr:
            raise MemoryError(f"bfq_host_alloc({nbytes})")
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr))[:self.nbytes]

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.L.bfq_host_free(self.ptr)
            self.ptr = None

    __del__ = free


class JobResult:
    """Outputs of Engine.fastq_job: arrays trimmed to their lengths + where every input part's share starts."""
    __slots__ = ("fastq", "dna", "qs", "hdr", "n_reads", "total_bases", "part_reads", "part_fastq_off",
                 "part_stream_off", "part_hdr_off", "stats")


def text_line_counts(buf, chunk=1 << 20, threads=0):
    """Number of newlines in every `chunk` bytes of a text (host threads, no GPU): uint64 array."""
    a = _u8(buf)
    nch = (len(a) + chunk - 1) // chunk
    counts = np.zeros(max(nch, 1), np.uint64)
    rc = _lib.lib().bfq_text_count_lines(_ptr(a) if len(a) else None, len(a), chunk, _ptr(counts), threads)
    if rc:
        raise BfqError(rc, "bfq_text_count_lines")
    return counts[:nch]


def text_nth_newline(buf, k):
    a = _u8(buf)
    return int(_lib.lib().bfq_text_nth_newline(_ptr(a) if len(a) else None, len(a), k))


def file_put(fd, data, offset, threads=0):
    """The bytes of `data` (uint8 array / bytes) into the open file `fd` at `offset` (bfq_file_put: several threads, no GPU)."""
    a = _u8(data)
    if len(a):
        rc = _lib.lib().bfq_file_put(fd, offset, _ptr(a), len(a), threads)
        if rc:
            raise BfqError(rc, "bfq_file_put")


class FileRange:
    """The byte range [offset, offset + nbytes) of an open file as a uint8 array whose pages are the file's (bfq_file_map:
    allocated, mapped and populated by a few threads): an output buffer that needs no writing afterwards.  .array is None
    when the file cannot be mapped.  close() unmaps."""

    def __init__(self, fd, offset, nbytes, threads=0):
        self.L = _lib.lib()
        self.offset, self.nbytes = int(offset), int(nbytes)
        self.ptr = self.L.bfq_file_map(fd, self.offset, self.nbytes, threads) if self.nbytes else None
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr)) if self.ptr else None

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.L.bfq_file_unmap(self.ptr, self.offset, self.nbytes)
            self.ptr = None

    __del__ = close


class HostText:
    """Host-side text helpers of libbfqhip.so (no GPU involved)."""
    FileRange = FileRange
    text_line_counts = staticmethod(text_line_counts)
    text_nth_newline = staticmethod(text_nth_newline)
    file_put = staticmethod(file_put)


class Engine:
    """One GPU context (one stream, one device workspace). Not thread-safe."""
    host = HostText

    def __init__(self, device=0, **params):
        self.L = _lib.lib()
        self.params = make_params(**params)
        self.h = self.L.bfq_create(device, C.byref(self.params))
        if not self.h:
            raise BfqError(-2, self.L.bfq_create_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.bfq_destroy(self.h)
            self.h = None

    __del__ = close

    def set_params(self, **params):
        self.params = make_params(**params)
        self._ck(self.L.bfq_set_params(self.h, C.byref(self.params)))

    def _ck(self, rc):
        if rc != 0:
            raise BfqError(rc, self.L.bfq_last_error(self.h).decode())

    # ---- step 1
    def build_ebwt(self, bases, quals, roff, term_out=ord("#")):
        bases = np.ascontiguousarray(bases, np.uint8); quals = np.ascontiguousarray(quals, np.uint8)
        roff = np.ascontiguousarray(roff, np.uint64)
        N = len(roff) - 1
        n = int(roff[-1]) + N
        bwt = np.empty(n, np.uint8); qs = np.empty(n, np.uint8); lcp = np.empty(n, np.uint16)
        self._ck(self.L.bfq_build_ebwt(self.h, _ptr(bases), _ptr(quals), _ptr(roff), N, term_out,
                                       _ptr(bwt), _ptr(qs), _ptr(lcp)))
        return bwt, qs, lcp

    # ---- steps 2-4
    def smooth_invert(self, bwt, qs, lcp=None, out=None):
        """out: optional (bases u8[n-N], quals u8[n-N], roff u64[N+1]) arrays to fill (e.g. pinned)."""
        bwt = np.ascontiguousarray(bwt, np.uint8); qs = np.ascontiguousarray(qs, np.uint8)
        n = len(bwt)
        N = C.c_uint64(0)
        self.L.bfq_count_reads(_ptr(bwt), n, self.params.term & 0xFF, C.byref(N))
        N = N.value
        lcp_bytes = 0
        if lcp is not None:
            lcp = np.ascontiguousarray(lcp)
            lcp_bytes = lcp.dtype.itemsize
        if out is not None:
            ob, oq, oroff = out
            assert len(ob) >= n - N and len(oq) >= n - N and len(oroff) >= N + 1 and oroff.dtype == np.uint64
        else:
            ob = np.empty(max(n - N, 1), np.uint8); oq = np.empty(max(n - N, 1), np.uint8)
            oroff = np.empty(N + 1, np.uint64)
        st = _lib.Stats()
        self._ck(self.L.bfq_smooth_invert(self.h, _ptr(bwt), _ptr(qs), _ptr(lcp), lcp_bytes, n,
                                          _ptr(ob), _ptr(oq), _ptr(oroff), C.byref(st)))
        return ob[:n - N], oq[:n - N], oroff[:N + 1], st.as_dict()

    # ---- fused
    def run_reads(self, bases, quals, roff):
        bases = np.ascontiguousarray(bases, np.uint8); quals = np.ascontiguousarray(quals, np.uint8)
        roff = np.ascontiguousarray(roff, np.uint64)
        N = len(roff) - 1
        ob = np.empty(max(len(bases), 1), np.uint8); oq = np.empty(max(len(bases), 1), np.uint8)
        st = _lib.Stats()
        self._ck(self.L.bfq_run_reads(self.h, _ptr(bases), _ptr(quals), _ptr(roff), N, _ptr(ob), _ptr(oq),
                                      C.byref(st)))
        return ob[:len(bases)], oq[:len(bases)], st.as_dict()

    def run_reads_device(self, d_bases, d_quals, d_roff, N, total, d_out_bases, d_out_quals):
        """All arguments are raw device pointers (ints), e.g. torch tensor .data_ptr()."""
        st = _lib.Stats()
        self._ck(self.L.bfq_run_reads_device(self.h, d_bases, d_quals, d_roff, N, total, d_out_bases,
                                             d_out_quals, C.byref(st)))
        return st.as_dict()

    # ---- FASTQ text in / out, parsed and formatted on the GPU
    def fastq_build_ebwt(self, text, term_out=ord("#"), want_lcp=True, out=None):
        """gsufsort / eGap on the bytes of a FASTQ file -> (bwt, qs, lcp16).  out: optional (bwt, qs, lcp16 or None) arrays."""
        buf = _u8(text)
        if out is not None:
            bwt, qs, lcp = out
            cap = min(len(bwt), len(qs), len(lcp) if lcp is not None else len(bwt))
            want_lcp = lcp is not None
        else:
            cap = len(buf) // 2 + 1
            bwt = np.empty(cap, np.uint8); qs = np.empty(cap, np.uint8)
            lcp = np.empty(cap, np.uint16) if want_lcp else None
        n = C.c_uint64(0); N = C.c_uint64(0)
        self._ck(self.L.bfq_fastq_build_ebwt(self.h, _ptr(buf), len(buf), term_out, _ptr(bwt), _ptr(qs), _ptr(lcp), cap,
                                             C.byref(n), C.byref(N)))
        return bwt[:n.value], qs[:n.value], (lcp[:n.value] if want_lcp else None)

    def fastq_run(self, text, keep_headers=False):
        """The whole path on the bytes of a FASTQ file -> (smoothed FASTQ bytes, stats)."""
        buf = np.frombuffer(text, np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
        out = np.empty(len(buf) + 16, np.uint8)
        ol = C.c_uint64(0)
        st = _lib.Stats()
        self._ck(self.L.bfq_fastq_run(self.h, _ptr(buf), len(buf), 1 if keep_headers else 0, _ptr(out), len(out),
                                      C.byref(ol), C.byref(st)))
        return out[:ol.value].tobytes(), st.as_dict()

    def fastq_run_streams(self, text, want_headers=True):
        """The whole path with the result as the streams of BFQzip.py --m2/--m3 (BFQzip.py:19-21,192-251):
        (OUT.fq.dna bytes, OUT.fq.qs bytes, OUT.h bytes or None, stats)."""
        buf = np.frombuffer(text, np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
        cap = len(buf) + 16
        dna = np.empty(cap, np.uint8); qs = np.empty(cap, np.uint8)
        hdr = np.empty(cap, np.uint8) if want_headers else None
        sl = C.c_uint64(0); hl = C.c_uint64(0)
        st = _lib.Stats()
        self._ck(self.L.bfq_fastq_run_streams(self.h, _ptr(buf), len(buf), _ptr(dna), _ptr(qs), cap, C.byref(sl),
                                              _ptr(hdr) if want_headers else None, cap, C.byref(hl), C.byref(st)))
        return (dna[:sl.value].tobytes(), qs[:sl.value].tobytes(),
                hdr[:hl.value].tobytes() if want_headers else None, st.as_dict())

    def fastq_job(self, parts, keep_headers=False, fastq=True, streams=False, hdr=False, out=None, compress=False):
        """One block of BFQzip_parallel.py in one call (bfq_fastq_run_job): `parts` = 1..4 byte ranges (bytes,
        uint8 arrays, memmap slices) processed as one collection; outputs as asked: the FASTQ text, the --m2
        streams (dna, qs), the --m3 header stream.  `out` may give reusable output arrays (e.g. PinnedBuffer.array)
        under the keys 'fastq', 'dna', 'qs', 'hdr'.  compress=True: the streams come back as BFQRANS2 containers
        (steps 1-5 of the reference in one call; stream_decompress gives the raw stream)."""
        arrs = [_u8(p) for p in parts]
        np_ = len(arrs)
        tp = (_lib.TextPart * np_)()
        for i, a in enumerate(arrs):
            tp[i].data = a.ctypes.data if len(a) else None
            tp[i].len = len(a)
        inlen = sum(len(a) for a in arrs)
        out = out or {}

        def buf(key, want, size):
            if not want:
                return None
            b = out.get(key)                      # a caller's buffer is used as it is (too small: BFQ_E_ARG)
            return b if b is not None else np.empty(size, np.uint8)
        J = _lib.FastqJob()
        J.parts = tp; J.nparts = np_; J.keep_headers = 1 if keep_headers else 0
        J.compress_streams = int(compress)                     # 0 raw, 1 containers of the streams, 2 eBWT-domain containers
        bf = buf("fastq", fastq, inlen + 5 * np_ + 16)
        zcap = (2 * int(self.L.bfq_stream_bound(inlen)) + 64) if compress else 0      # a tiny stream's container is larger than the stream
        bd, bq = buf("dna", streams, max(inlen + 16, zcap)), buf("qs", streams, max(inlen + 16, zcap))
        bh = buf("hdr", hdr, max(inlen + 16, zcap))
        if bf is not None:
            J.out_fastq = bf.ctypes.data; J.cap_fastq = len(bf)
        if bd is not None:
            J.out_dna = bd.ctypes.data; J.out_qs = bq.ctypes.data; J.cap_stream = min(len(bd), len(bq))
        if bh is not None:
            J.out_hdr = bh.ctypes.data; J.cap_hdr = len(bh)
        st = _lib.Stats()
        self._ck(self.L.bfq_fastq_run_job(self.h, C.byref(J), C.byref(st)))
        r = JobResult()
        r.fastq = bf[:J.fastq_len] if bf is not None else None
        r.dna = bd[:J.dna_bytes] if bd is not None else None
        r.qs = bq[:J.qs_bytes] if bq is not None else None
        r.hdr = bh[:J.hdr_bytes] if bh is not None else None
        r.n_reads, r.total_bases = int(J.n_reads), int(J.total_bases)
        r.part_reads = [int(J.part_reads[i]) for i in range(np_ + 1)]
        r.part_fastq_off = [int(J.part_fastq_off[i]) for i in range(np_ + 1)]
        r.part_stream_off = [int(J.part_stream_off[i]) for i in range(np_ + 1)]
        r.part_hdr_off = [int(J.part_hdr_off[i]) for i in range(np_ + 1)]
        r.stats = st.as_dict()
        return r

    # ---- global mode (one collection over several GPUs, unsharded result): the per-GPU pieces; tensors are torch
    #      uint8 tensors on this engine's device (bfqzip_amd/parallel.py run_global drives them)
    tensor_device = "cuda"

    def glob_begin(self, parts):
        arrs = [_u8(p) for p in parts]
        tp = (_lib.TextPart * max(len(arrs), 1))()
        for i, a in enumerate(arrs):
            tp[i].data = a.ctypes.data if len(a) else None
            tp[i].len = len(a)
        N = (C.c_uint64 * len(arrs))(); T = (C.c_uint64 * len(arrs))()
        self._ck(self.L.bfq_glob_begin(self.h, tp, len(arrs), N, T))
        return [int(x) for x in N], [int(x) for x in T]                  # reads / bases of every part

    def glob_local_text(self, t8, q8):
        self._ck(self.L.bfq_glob_local_text(self.h, t8.data_ptr(), q8.data_ptr()))

    def glob_pile_counts(self, t8, n):
        cnt = np.zeros(36, np.uint64)
        self._ck(self.L.bfq_glob_pile_counts(self.h, t8.data_ptr() if n else None, n, _ptr(cnt)))
        return cnt.reshape(6, 6)

    def glob_init_out(self, t8, q8, n, sym, qual):
        self._ck(self.L.bfq_glob_init_out(self.h, t8.data_ptr(), q8.data_ptr(), n, sym.data_ptr(), qual.data_ptr()))

    def glob_run_pile(self, t8, q8, n, s, s2, sym, qual):
        st = _lib.Stats()
        self._ck(self.L.bfq_glob_run_pile(self.h, t8.data_ptr(), q8.data_ptr(), n, s, s2, sym.data_ptr(), qual.data_ptr(), C.byref(st)))
        return st.as_dict()

    def glob_finish(self, dna, qs, keep_headers=False, fastq=True, streams=False, hdr=False, text_len=0, nparts=1, out=None):
        """dna / qs: this block's line streams (torch uint8, device).  Returns a JobResult like fastq_job.  `out`: reusable
        output arrays under 'fastq', 'dna', 'qs', 'hdr' (e.g. PinnedBuffer.array: direct DMA), used when large enough."""
        J = _lib.FastqJob()
        J.nparts = 0; J.keep_headers = 1 if keep_headers else 0
        sl = int(dna.numel())
        out = out or {}

        def buf(key, want, need, slack):
            if not want:
                return None
            b = out.get(key)                     # a caller's buffer of the exact size will do (e.g. a mapped file range)
            return b if (b is not None and len(b) >= need) else np.empty(need + slack, np.uint8)
        bf = buf("fastq", fastq, text_len, 32)
        bd = buf("dna", streams, sl, 16)
        bq = buf("qs", streams, sl, 16)
        bh = buf("hdr", hdr, text_len, 16)
        if bf is not None:
            J.out_fastq = bf.ctypes.data; J.cap_fastq = len(bf)
        if bd is not None:
            J.out_dna = bd.ctypes.data; J.out_qs = bq.ctypes.data; J.cap_stream = len(bd)
        if bh is not None:
            J.out_hdr = bh.ctypes.data; J.cap_hdr = len(bh)
        self._ck(self.L.bfq_glob_finish(self.h, dna.data_ptr() if sl else None, qs.data_ptr() if sl else None, C.byref(J)))
        r = JobResult()
        r.fastq = bf[:J.fastq_len] if bf is not None else None
        r.dna = bd[:J.dna_bytes] if bd is not None else None
        r.qs = bq[:J.qs_bytes] if bq is not None else None
        r.hdr = bh[:J.hdr_bytes] if bh is not None else None
        r.n_reads, r.total_bases = int(J.n_reads), int(J.total_bases)
        r.part_reads = [int(J.part_reads[i]) for i in range(nparts + 1)]
        r.part_fastq_off = [int(J.part_fastq_off[i]) for i in range(nparts + 1)]
        r.part_stream_off = [int(J.part_stream_off[i]) for i in range(nparts + 1)]
        r.part_hdr_off = [int(J.part_hdr_off[i]) for i in range(nparts + 1)]
        r.stats = {}
        return r

    def smooth_invert_fastq(self, bwt, qs, lcp=None, headers=None):
        """bfq_int / bfq_ext writing the FASTQ text; headers = bytes of the -H file or None."""
        bwt = np.ascontiguousarray(bwt, np.uint8); qs = np.ascontiguousarray(qs, np.uint8)
        n = len(bwt)
        N = C.c_uint64(0)
        self.L.bfq_count_reads(_ptr(bwt), n, self.params.term & 0xFF, C.byref(N))
        N = N.value
        lcp_bytes = 0
        if lcp is not None:
            lcp = np.ascontiguousarray(lcp); lcp_bytes = lcp.dtype.itemsize
        hb = np.frombuffer(headers, np.uint8) if headers is not None else None
        cap = int(self.L.bfq_fastq_out_bound(n - N, N, len(hb) if hb is not None else 0)) + 16
        out = np.empty(cap, np.uint8)
        ol = C.c_uint64(0)
        st = _lib.Stats()
        self._ck(self.L.bfq_smooth_invert_fastq(self.h, _ptr(bwt), _ptr(qs), _ptr(lcp), lcp_bytes, n, _ptr(hb),
                                                len(hb) if hb is not None else 0, _ptr(out), cap, C.byref(ol), C.byref(st)))
        return out[:ol.value].tobytes(), st.as_dict()

    def fetch_ebwt(self, n, out=None):
        bwt, qs, lcp = out if out is not None else (np.empty(n, np.uint8), np.empty(n, np.uint8), np.empty(n, np.uint16))
        self._ck(self.L.bfq_fetch_ebwt(self.h, _ptr(bwt), _ptr(qs), _ptr(lcp)))
        return bwt, qs, lcp

    # ---- synthetic reads
    def synth_device(self, spec, d_bases, d_quals, d_roff):
        self._ck(self.L.bfq_synth_device(self.h, C.byref(spec), d_bases, d_quals, d_roff))

    def synth_fastq(self, spec, out):
        """The synthetic reads of `spec` as FASTQ text (headers "@SYN.<n>") into the uint8 array `out`; returns the length."""
        ol = C.c_uint64(0)
        self._ck(self.L.bfq_synth_fastq(self.h, C.byref(spec), _ptr(out), len(out), C.byref(ol)))
        return int(ol.value)

    # ---- stream codec (step 5 of the reference: BFQzip.py:253-275)
    def stream_compress(self, data, out=None):
        """BFQRANS2 container of the bytes `data` (uint8 array); returns a uint8 array (a view of `out` when given)."""
        data = _u8(data)
        cap = int(self.L.bfq_stream_bound(len(data)))
        if out is None:
            out = np.empty(cap, np.uint8)
        ol = C.c_uint64(0)
        self._ck(self.L.bfq_stream_compress(self.h, _ptr(data), len(data), _ptr(out), len(out), C.byref(ol)))
        return out[:int(ol.value)]

    def stream_decompress(self, blob, out=None):
        """The raw bytes of a BFQRANS2 container."""
        blob = _u8(blob)
        n = int(self.L.bfq_stream_raw_len(_ptr(blob), len(blob)))
        if n < 0:
            raise BfqError(-1, "not a BFQRANS2 / BFQDNAC1 / BFQLINE1 stream")
        if out is None:
            out = np.empty(max(n, 1), np.uint8)
        ol = C.c_uint64(0)
        self._ck(self.L.bfq_stream_decompress(self.h, _ptr(blob), len(blob), _ptr(out), len(out), C.byref(ol)))
        return out[:int(ol.value)]

    def ebwt_decode(self, bwtz, qsz, out=None):
        """The line streams (dna, qs) of a pair of eBWT-domain containers (fastq_job(compress=2)); returns (dna, qs, n_reads)."""
        bwtz, qsz = _u8(bwtz), _u8(qsz)
        if len(bwtz) < 32 or bytes(bwtz[:8]) != b"BFQEBWT1":
            raise BfqError(-1, "not a BFQEBWT1 stream")
        n = int(np.frombuffer(bwtz[8:16].tobytes(), np.uint64)[0])
        dna, qs = out if out is not None else (np.empty(n + 16, np.uint8), np.empty(n + 16, np.uint8))
        sl, nr = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.L.bfq_stream_ebwt_decode(self.h, _ptr(bwtz), len(bwtz), _ptr(qsz), len(qsz), _ptr(dna), _ptr(qs),
                                               min(len(dna), len(qs)), C.byref(sl), C.byref(nr)))
        return dna[:int(sl.value)], qs[:int(sl.value)], int(nr.value)

    def stream_compress_device(self, d_in, n, d_out, cap):
        """Device-resident form (after stream_reserve(n)); returns the container's length."""
        ol = C.c_uint64(0)
        self._ck(self.L.bfq_stream_compress_device(self.h, d_in, n, d_out, cap, C.byref(ol)))
        return int(ol.value)

    def stream_reserve(self, n):
        self._ck(self.L.bfq_stream_reserve(self.h, n))

    def stream_bound(self, n):
        return int(self.L.bfq_stream_bound(n))

    # ---- profiling
    def prof_reset(self):
        self.L.bfq_prof_reset(self.h)

    def prof(self):
        out = {}
        name = C.create_string_buffer(64)
        for i in range(self.L.bfq_prof_count(self.h)):
            ms = C.c_double(); ln = C.c_uint64(); by = C.c_double()
            self.L.bfq_prof_get(self.h, i, name, 64, C.byref(ms), C.byref(ln), C.byref(by))
            if ln.value:
                out[name.value.decode()] = {"ms": ms.value, "launches": ln.value, "alg_bytes": by.value}
        return out

    def prof_trace_select(self, kernel):
        """Keep the per-launch durations of `kernel` (a name prof() reports, or None) from now on."""
        name = C.create_string_buffer(64)
        idx = -1
        for i in range(self.L.bfq_prof_count(self.h)):
            self.L.bfq_prof_get(self.h, i, name, 64, None, None, None)
            if kernel is not None and name.value.decode() == kernel:
                idx = i
        if kernel is not None and idx < 0:
            raise KeyError(kernel)
        self._ck(self.L.bfq_prof_trace_select(self.h, idx))

    def prof_trace(self):
        """Per-launch milliseconds of the selected kernel since the last prof_reset(), launch order (float32 array)."""
        k = int(self.L.bfq_prof_trace(self.h, None, 0))
        out = np.empty(max(k, 0), np.float32)
        if k > 0:
            self.L.bfq_prof_trace(self.h, _ptr(out), k)
        return out

    def workspace_bytes(self):
        return int(self.L.bfq_workspace_bytes(self.h))


def synth_spec(N, L, Lmax=None, seed=20240807, **kw):
    s = _lib.Synth()
    _lib.lib().bfq_synth_default(C.byref(s), N, L)
    s.seed = seed
    if Lmax is not None:
        s.Lmax = Lmax
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def synth_host(spec):
    """Generate the synthetic reads on the host (same bytes as the device generator)."""
    L = _lib.lib()
    total = int(L.bfq_synth_total(C.byref(spec)))
    bases = np.empty(max(total, 1), np.uint8); quals = np.empty(max(total, 1), np.uint8)
    roff = np.empty(spec.N + 1, np.uint64)
    rc = L.bfq_synth_host(C.byref(spec), _ptr(bases), _ptr(quals), _ptr(roff))
    if rc:
        raise BfqError(rc, "bfq_synth_host")
    return bases[:total], quals[:total], roff
