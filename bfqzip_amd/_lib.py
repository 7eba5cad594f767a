"""ctypes loader for libbfqhip.so (the HIP kernels + C-ABI of include/bfqzip_hip.h).

There is no CPU fallback: if the library is missing it is built (hipcc), and
if it cannot be loaded or no GPU is present the caller gets an exception.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbfqhip.so")
_LIB = None

SYMBOLS = [
    "bfq_default_params", "bfq_create", "bfq_create_error", "bfq_destroy", "bfq_set_params",
    "bfq_last_error", "bfq_stream", "bfq_device_count", "bfq_pick_device", "bfq_device_lease", "bfq_device_release",
    "bfq_phase_enable", "bfq_phase", "bfq_phase_report", "bfq_output_prefault", "bfq_fastq_rows_estimate", "bfq_build_ebwt", "bfq_count_reads",
    "bfq_smooth_invert", "bfq_run_reads", "bfq_run_reads_device", "bfq_fetch_ebwt",
    "bfq_fastq_out_bound", "bfq_fastq_build_ebwt", "bfq_fastq_run", "bfq_fastq_run_streams",
    "bfq_smooth_invert_fastq", "bfq_fastq_run_job", "bfq_host_alloc", "bfq_host_free",
    "bfq_text_count_lines", "bfq_text_nth_newline", "bfq_file_put", "bfq_file_map", "bfq_file_unmap", "bfq_fastq_build_ebwt_fd", "bfq_smooth_invert_fastq_fd",
    "bfq_glob_begin", "bfq_glob_local_text", "bfq_glob_pile_counts", "bfq_glob_init_out", "bfq_glob_run_pile", "bfq_glob_finish",
    "bfq_synth_default", "bfq_synth_total", "bfq_synth_host", "bfq_synth_device", "bfq_synth_fastq",
    "bfq_prof_enable", "bfq_prof_reset", "bfq_prof_count", "bfq_prof_get", "bfq_prof_trace_select", "bfq_prof_trace",
    "bfq_stream_bound", "bfq_stream_raw_len", "bfq_stream_compress", "bfq_stream_decompress",
    "bfq_stream_reserve", "bfq_stream_compress_device", "bfq_stream_ebwt_decode",
    "bfq_workspace_bytes", "bfq_version",
]


class Params(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("K", "m", "v", "f", "t", "term", "M", "B", "ext", "piles", "ws_cap_mib")] + \
               [("reserved", C.c_int32 * 5)]


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "num_clust", "num_clust_discarded", "num_clust_amb_discarded", "num_clust_mod",
        "num_clust_alleq", "bases_inside", "qs_smoothed", "modified",
        "n_rows", "n_reads", "n_segments", "n_big_segments")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Synth(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("N", C.c_uint64), ("Lmin", C.c_uint32), ("Lmax", C.c_uint32),
                ("coverage", C.c_uint32), ("err_ppm", C.c_uint32), ("n_ppm", C.c_uint32),
                ("snp_every", C.c_uint32), ("dsnp_every", C.c_uint32), ("both_strands", C.c_uint32),
                ("first", C.c_uint64), ("collection", C.c_uint64), ("reserved", C.c_uint32 * 2)]


MAX_PARTS = 4


class TextPart(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_uint64)]


class FastqJob(C.Structure):
    _fields_ = [("parts", C.POINTER(TextPart)), ("nparts", C.c_int32), ("keep_headers", C.c_int32),
                ("out_fastq", C.c_void_p), ("cap_fastq", C.c_uint64),
                ("out_dna", C.c_void_p), ("out_qs", C.c_void_p), ("cap_stream", C.c_uint64),
                ("out_hdr", C.c_void_p), ("cap_hdr", C.c_uint64),
                ("fastq_len", C.c_uint64), ("stream_len", C.c_uint64), ("hdr_len", C.c_uint64),
                ("n_reads", C.c_uint64), ("total_bases", C.c_uint64),
                ("part_reads", C.c_uint64 * (MAX_PARTS + 1)), ("part_fastq_off", C.c_uint64 * (MAX_PARTS + 1)),
                ("part_stream_off", C.c_uint64 * (MAX_PARTS + 1)), ("part_hdr_off", C.c_uint64 * (MAX_PARTS + 1)),
                ("compress_streams", C.c_int32), ("reserved0", C.c_int32),
                ("dna_bytes", C.c_uint64), ("qs_bytes", C.c_uint64), ("hdr_bytes", C.c_uint64)]


def build(clean=False):
    """Compile libbfqhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    d = os.path.join(_HERE, "csrc")
    if clean:
        subprocess.check_call(["make", "-s", "-C", d, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", d, "all"])


def lib():
    global _LIB
    if _LIB is None:
        # torch bundles its own HIP/HSA runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).
        # Two HIP runtimes in one process cannot both own the GPU, so when torch is installed it
        # is imported first and libbfqhip.so binds to the runtime torch already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.bfq_create.restype = C.c_void_p
        L.bfq_create.argtypes = [C.c_int, C.POINTER(Params)]
        L.bfq_create_error.restype = C.c_char_p
        L.bfq_destroy.argtypes = [C.c_void_p]
        L.bfq_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.bfq_last_error.restype = C.c_char_p
        L.bfq_last_error.argtypes = [C.c_void_p]
        L.bfq_stream.restype = C.c_void_p
        L.bfq_stream.argtypes = [C.c_void_p]
        L.bfq_version.restype = C.c_char_p
        L.bfq_workspace_bytes.restype = C.c_uint64
        L.bfq_workspace_bytes.argtypes = [C.c_void_p]
        L.bfq_synth_total.restype = C.c_uint64
        vp, u64 = C.c_void_p, C.c_uint64
        L.bfq_pick_device.argtypes = [C.c_char_p, C.c_int]
        L.bfq_device_lease.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double)]
        L.bfq_device_release.argtypes = [C.c_int]
        L.bfq_phase_enable.argtypes = [C.c_int]
        L.bfq_phase.argtypes = [C.c_char_p]
        L.bfq_phase_report.argtypes = [C.c_char_p]
        L.bfq_output_prefault.argtypes = [C.c_int, u64, u64]
        L.bfq_fastq_rows_estimate.restype = u64
        L.bfq_fastq_rows_estimate.argtypes = [C.c_int, u64]
        L.bfq_build_ebwt.argtypes = [vp, vp, vp, vp, u64, C.c_int, vp, vp, vp]
        L.bfq_count_reads.argtypes = [vp, u64, C.c_int, C.POINTER(u64)]
        L.bfq_smooth_invert.argtypes = [vp, vp, vp, vp, C.c_int, u64, vp, vp, vp, C.POINTER(Stats)]
        L.bfq_run_reads.argtypes = [vp, vp, vp, vp, u64, vp, vp, C.POINTER(Stats)]
        L.bfq_run_reads_device.argtypes = [vp, vp, vp, vp, u64, u64, vp, vp, C.POINTER(Stats)]
        L.bfq_fetch_ebwt.argtypes = [vp, vp, vp, vp]
        L.bfq_fastq_out_bound.restype = C.c_uint64
        L.bfq_fastq_out_bound.argtypes = [u64, u64, u64]
        L.bfq_fastq_build_ebwt.argtypes = [vp, vp, u64, C.c_int, vp, vp, vp, u64, C.POINTER(u64), C.POINTER(u64)]
        L.bfq_fastq_run.argtypes = [vp, vp, u64, C.c_int, vp, u64, C.POINTER(u64), C.POINTER(Stats)]
        L.bfq_fastq_run_streams.argtypes = [vp, vp, u64, vp, vp, u64, C.POINTER(u64), vp, u64, C.POINTER(u64),
                                            C.POINTER(Stats)]
        L.bfq_smooth_invert_fastq.argtypes = [vp, vp, vp, vp, C.c_int, u64, vp, u64, vp, u64, C.POINTER(u64),
                                              C.POINTER(Stats)]
        L.bfq_fastq_run_job.argtypes = [vp, C.POINTER(FastqJob), C.POINTER(Stats)]
        L.bfq_glob_begin.argtypes = [vp, C.POINTER(TextPart), C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.bfq_glob_local_text.argtypes = [vp, vp, vp]
        L.bfq_glob_pile_counts.argtypes = [vp, vp, u64, vp]
        L.bfq_glob_init_out.argtypes = [vp, vp, vp, u64, vp, vp]
        L.bfq_glob_run_pile.argtypes = [vp, vp, vp, u64, C.c_int, C.c_int, vp, vp, C.POINTER(Stats)]
        L.bfq_glob_finish.argtypes = [vp, vp, vp, C.POINTER(FastqJob)]
        L.bfq_host_alloc.restype = vp
        L.bfq_host_alloc.argtypes = [u64]
        L.bfq_host_free.argtypes = [vp]
        L.bfq_text_count_lines.argtypes = [vp, u64, u64, vp, C.c_int]
        L.bfq_file_put.argtypes = [C.c_int, u64, vp, u64, C.c_int]
        L.bfq_file_map.restype = vp
        L.bfq_file_map.argtypes = [C.c_int, u64, u64, C.c_int]
        L.bfq_file_unmap.argtypes = [vp, u64, u64]
        L.bfq_text_nth_newline.restype = C.c_int64
        L.bfq_text_nth_newline.argtypes = [vp, u64, u64]
        L.bfq_synth_default.argtypes = [C.POINTER(Synth), u64, C.c_uint32]
        L.bfq_synth_total.argtypes = [C.POINTER(Synth)]
        L.bfq_synth_host.argtypes = [C.POINTER(Synth), vp, vp, vp]
        L.bfq_synth_device.argtypes = [vp, C.POINTER(Synth), vp, vp, vp]
        L.bfq_synth_fastq.argtypes = [vp, C.POINTER(Synth), vp, u64, C.POINTER(u64)]
        L.bfq_stream_bound.restype = u64
        L.bfq_stream_bound.argtypes = [u64]
        L.bfq_stream_raw_len.restype = C.c_int64
        L.bfq_stream_raw_len.argtypes = [vp, u64]
        L.bfq_stream_compress.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
        L.bfq_stream_decompress.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
        L.bfq_stream_reserve.argtypes = [vp, u64]
        L.bfq_stream_ebwt_decode.argtypes = [vp, vp, u64, vp, u64, vp, vp, u64, C.POINTER(u64), C.POINTER(u64)]
        L.bfq_stream_compress_device.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
        L.bfq_prof_enable.argtypes = [vp, C.c_int]
        L.bfq_prof_reset.argtypes = [vp]
        L.bfq_prof_count.argtypes = [vp]
        L.bfq_prof_trace_select.argtypes = [vp, C.c_int]
        L.bfq_prof_trace.restype = C.c_int64
        L.bfq_prof_trace.argtypes = [vp, vp, u64]
        L.bfq_prof_get.argtypes = [vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double),
                                   C.POINTER(u64), C.POINTER(C.c_double)]
        _LIB = L
    return _LIB
