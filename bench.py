#!/usr/bin/env python3
"""bench.py -- Mbases/s of the hot path (eBWT build -> clusters/smoothing -> LF inversion)
on synthetic short reads, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 30Mx150|1Mx100|<reads>x<len>]
                  [--scaling weak|strong]

A step = one pass of the whole path over one block of reads resident in HBM
(bfq_run_reads_device): `value`.  Blocks never interact (the BFQzip_parallel.py split), so there is
no data-path collective; only block sizes are exchanged.
  --scaling weak   (default) every rank owns its own block of the named workload;
  --scaling strong = BASELINE.json configs[3]: ONE collection of the named size cut by
                   split_blocks(reads, N) (BFQzip_parallel.py:288-323), rank r runs block r.
                   N = 1: the same run as weak.
Rank 0 prints ONE JSON line.  Besides the contract's fields it carries (N = 1 only):
  e2e_host      SURVEY 8(d)'s metric: FASTQ text in pinned host memory -> bfq_fastq_run_job ->
                the --m3 streams in pinned host memory, PCIe transfers included;
  dropin_wall_s the drop-in executables as BFQzip.py runs them (gsufsort, then bfq_int) on
                /dev/shm files: wall seconds per tool;
  cpu_baseline  the reference CPU path on a bounded sample (1 core), and sample_parity;
and for N > 1 in weak mode `strong`: the configs[3] measurement taken after the timed region.
"""
import argparse, json, os, shutil, subprocess, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def alg_bytes_per_base(L):
    """A(L) = (L+1)/L * (352.7 + 3(L+1)/16) bytes per base."""
    return (L + 1) / L * (352.7 + 3 * (L + 1) / 16)


def pmc_traffic(kernel, rows):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*/traffic_per_row.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same bench command; FETCH doubled for the
    coalesced streaming kernels as MI355X_MICROARCH.md prescribes, raw for random sector reads).  None if absent."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic_per_row.json"))):
        try:
            best = json.load(open(p))
        except Exception:
            pass
    ks = best.get("kernels", {}) if best else {}
    name = kernel if kernel in ks else next((x for x in ks if x.split("<")[0] == kernel), None)
    if name is None:
        return None
    k = ks[name]
    streaming = kernel not in ("k_invert", "k_invert<1>", "k_invert<0>", "k_refine_chunk", "k_cluster")
    fetch = k["fetch_B_per_row_raw"] * (2.0 if streaming else 1.0)
    return round((fetch + k["write_B_per_row"]) * rows)


def pmc_traffic_per_row():
    """Sum over the step's kernels of the PMC-measured HBM bytes per row (the newest profiles/*/traffic_per_row.json)."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic_per_row.json"))):
        try:
            best = json.load(open(p))
        except Exception:
            pass
    if not best:
        return None
    if "step_total_B_per_row" in best:
        return best["step_total_B_per_row"]
    ks = best.get("kernels", {})
    steps = ks.get("k_refine_chunk", {}).get("calls")
    if not steps:
        return None
    tot = 0.0
    for name, k in ks.items():
        streaming = name.split("<")[0] not in ("k_invert", "k_refine_chunk", "k_cluster")
        tot += (k["fetch_B_per_row_raw"] * (2.0 if streaming else 1.0) + k["write_B_per_row"]) * k["calls"] / steps
    return tot


def parse_workload(w):
    a, b = w.lower().split("x")
    mult = 1
    if a.endswith("m"):
        a, mult = a[:-1], 1_000_000
    elif a.endswith("k"):
        a, mult = a[:-1], 1_000
    return int(float(a) * mult), int(b)


def cpu_baseline(api, orc, L, seed, sample_reads, params):
    """Reference CPU path on a bounded sample of the same workload, 1 core.
    step 1: the oracle's suffix sorter (stands in for gsufsort, an absent submodule);
    steps 2-4: the reference bfq_int compiled from /root/reference (oracle/_ref) if
    present ("reference"), else the oracle restatement ("port")."""
    from bfqzip_amd import fastq
    sp = api.synth_spec(sample_reads, L, seed=seed)
    b, q, r = api.synth_host(sp)
    t0 = time.perf_counter()
    bwt, qs, lcp = orc.build_ebwt(b, q, r)
    t1 = time.perf_counter()
    kind = "port"
    ref = orc.ref_binary(params["M"], params["B"])
    out = None
    if ref is not None:
        d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            bwt.tofile(d + "/x.bwt"); qs.tofile(d + "/x.bwt.qs")
            t2 = time.perf_counter()
            subprocess.check_call([ref, "-e", d + "/x.bwt", "-q", d + "/x.bwt.qs", "-o", d + "/o.fq", "-m", str(params["m"])],
                                  stdout=subprocess.DEVNULL, timeout=1200)
            t3 = time.perf_counter()
            out = open(d + "/o.fq", "rb").read()
            kind = "reference"
        except Exception:
            out = None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if out is None:
        t2 = time.perf_counter()
        ob, oq, oroff, st = orc.smooth_invert(bwt, qs, lcp.astype(np.uint32), orc.params(m=params["m"], M=params["M"], B=params["B"]))
        t3 = time.perf_counter()
        out = fastq.format_fastq(ob, oq, oroff)
    secs = (t1 - t0) + (t3 - t2)
    return {"value": len(b) / 1e6 / secs, "unit": "Mbases/s", "cores": 1, "kind": kind,
            "sample": f"{sample_reads}x{L} synthetic reads of the same generator; step 1 by the oracle's suffix sorter "
                      f"(stand-in for gsufsort, absent) {t1 - t0:.2f}s + steps 2-4 {t3 - t2:.2f}s"}, (b, q, r, out)


def e2e_host(api, eng, sp, N, L, iters, log):
    """FASTQ bytes in (pinned) host memory -> streams in (pinned) host memory, one bfq_fastq_run_job call."""
    cap = N * (2 * L + 30) + 4096
    pin = api.PinnedBuffer(cap)
    tlen = eng.synth_fastq(sp, pin.array)
    outs = {k: api.PinnedBuffer(N * (L + 1) + 4096 if k != "hdr" else N * 24 + 4096) for k in ("dna", "qs", "hdr")}
    ob = {k: v.array for k, v in outs.items()}
    text = pin.array[:tlen]
    res = eng.fastq_job([text], streams=True, hdr=True, fastq=False, out=ob)        # warm-up: sizes the workspace
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        res = eng.fastq_job([text], streams=True, hdr=True, fastq=False, out=ob)
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    log(f"e2e_host: {[round(t * 1e3) for t in ts]} ms")
    out = {"value": round(N * L / 1e6 / dt, 2), "unit": "Mbases/s", "ms": round(dt * 1e3, 2), "iters": iters,
           "bytes_in": int(tlen), "bytes_out": int(2 * len(res.dna) + len(res.hdr)),
           "what": "FASTQ text (headers '@SYN.<n>') in pinned host memory -> bfq_fastq_run_job -> OUT.fq.dna + OUT.fq.qs + OUT.h "
                   "in pinned host memory (H2D, GPU parse, whole path, GPU format, D2H); best of iters"}
    return out, pin, tlen, outs


def stream_codec(api, eng, outs, lens, N, L, log, with_cpu, text=None):
    """the step the reference hands to 7z / bsc (BFQzip.py:253-275) -- the three output streams of the run
    above through the GPU codec (bfq_stream_compress: pinned host buffer in, pinned host buffer out), sizes and wall time;
    with_cpu: bzip2 -9 / xz -2 on the first 32 MB of each stream beside it (the reference's own tools are not in its tree)."""
    import subprocess, tempfile
    res = {}
    cap = eng.stream_bound(max(lens.values()))
    pout = api.PinnedBuffer(cap)
    pback = api.PinnedBuffer(max(lens.values()) + 64)
    tot_raw = tot_cmp = 0
    tot_t = 0.0
    for k in ("dna", "qs", "hdr"):
        raw = outs[k].array[:lens[k]]
        eng.stream_compress(raw, out=pout.array)                 # warm-up: sizes the workspace
        eng.prof_reset()
        t0 = time.perf_counter()
        blob = eng.stream_compress(raw, out=pout.array)
        dt = time.perf_counter() - t0
        pr = eng.prof()
        t1 = time.perf_counter()
        back = eng.stream_decompress(blob, out=pback.array)
        dt2 = time.perf_counter() - t1
        ok = bool(len(back) == len(raw) and np.array_equal(back, raw))     # the whole stream (the container's checksum was verified by the decoder as well)
        res[k] = {"raw_bytes": int(len(raw)), "compressed_bytes": int(len(blob)), "bits_per_symbol": round(8.0 * len(blob) / max(len(raw), 1), 4),
                  "compress_wall_ms": round(dt * 1e3, 1), "decompress_wall_ms": round(dt2 * 1e3, 1), "round_trip_ok": ok,
                  "kernel_ms": round(pr.get("k_codec", {}).get("ms", 0.0), 1)}
        tot_raw += len(raw); tot_cmp += len(blob); tot_t += dt
        log(f"stream_codec {k}: {len(raw)} -> {len(blob)} bytes, {dt * 1e3:.0f} ms")
        # the DNA stream also on its first 256 MB: the first 32 MB of a 30x collection of a 150 Mbase genome are 0.2x coverage --
        # nothing a context model could learn from (there the static container is chosen); 256 MB are 1.8x
        for mb in ((32, 256) if k == "dna" else (32,)) if with_cpu else ():
            n = min(len(raw), mb << 20)
            nlp = np.flatnonzero(raw[max(0, n - 65536):n] == 10)       # whole lines (the line-delta transform wants them)
            if len(nlp):
                n = max(0, n - 65536) + int(nlp[-1]) + 1
            with tempfile.NamedTemporaryFile(dir="/dev/shm", delete=False) as f:
                f.write(raw[:n].tobytes())
            try:
                ref = {}
                for tool, cmd in (("bzip2 -9", ["bzip2", "-9", "-c", f.name]), ("xz -2", ["xz", "-2", "-T1", "-c", f.name]))[:2 if mb == 32 else 1]:
                    t0 = time.perf_counter()
                    o = subprocess.run(cmd, stdout=subprocess.PIPE, check=True).stdout
                    ref[tool] = {"bits_per_symbol": round(8.0 * len(o) / n, 4), "MB_per_s_1_core": round(n / 1e6 / (time.perf_counter() - t0), 1)}
                ours = eng.stream_compress(raw[:n])
                ref["this codec, same sample"] = {"bits_per_symbol": round(8.0 * len(ours) / n, 4), "container": bytes(ours[:8]).decode()}
                res[k][f"cpu_tools_{mb}MB_sample"] = ref
            finally:
                os.unlink(f.name)
    pout.free(); pback.free()
    if text is not None:                                         # steps 1-5 in ONE call: FASTQ text in, three containers out
        zo = {k: api.PinnedBuffer(eng.stream_bound(lens[k]) if lens[k] < (1 << 28) else lens[k]) for k in ("dna", "qs", "hdr")}
        zb = {k: v.array for k, v in zo.items()}
        eng.fastq_job([text], streams=True, hdr=True, fastq=False, out=zb, compress=True)
        t0 = time.perf_counter()
        z = eng.fastq_job([text], streams=True, hdr=True, fastq=False, out=zb, compress=True)
        dt = time.perf_counter() - t0
        same = all(int(len(getattr(z, k))) == res[k]["compressed_bytes"] for k in ("dna", "qs", "hdr"))
        res["fused_steps_1_to_5"] = {"wall_ms": round(dt * 1e3, 1), "Mbases_per_s": round(N * L / 1e6 / dt, 1),
                                     "bytes_in": int(len(text)), "bytes_out": int(len(z.dna) + len(z.qs) + len(z.hdr)),
                                     "containers_equal_separate_run": bool(same),
                                     "what": "bfq_fastq_run_job with compress_streams: FASTQ text (pinned) -> parse, eBWT, clusters, inversion, "
                                             "entropy coding, all on the GPU -> three containers (pinned: BFQDNAC1 / BFQRANS2 / BFQLINE1); the raw streams never cross the bus"}
        log(f"fused steps 1-5: {dt * 1e3:.0f} ms")
        # the same with eBWT-domain containers: rows of the edited eBWT instead of reads (no inversion on the compressing side)
        eng.fastq_job([text], streams=True, hdr=True, fastq=False, out=zb, compress=2)
        t0 = time.perf_counter()
        z2 = eng.fastq_job([text], streams=True, hdr=True, fastq=False, out=zb, compress=2)
        dt = time.perf_counter() - t0
        nb = int(len(z2.dna) + len(z2.qs) + len(z2.hdr))
        back = (api.PinnedBuffer(lens["dna"] + 64), api.PinnedBuffer(lens["qs"] + 64))
        t0 = time.perf_counter()
        d2, q2, nr = eng.ebwt_decode(z2.dna, z2.qs, out=(back[0].array, back[1].array))
        dt2 = time.perf_counter() - t0
        same2 = bool(nr == N and np.array_equal(d2, outs["dna"].array[:lens["dna"]]) and np.array_equal(q2, outs["qs"].array[:lens["qs"]]))
        res["ebwt_domain"] = {"wall_ms": round(dt * 1e3, 1), "Mbases_per_s": round(N * L / 1e6 / dt, 1), "bytes_out": nb,
                              "dna_container_bytes": int(len(z2.dna)), "qs_container_bytes": int(len(z2.qs)),
                              "ratio_to_raw_streams": round(tot_raw / max(nb, 1), 2), "decode_wall_ms": round(dt2 * 1e3, 1),
                              "decoded_streams_equal_e2e_host": same2,
                              "what": "bfq_fastq_run_job with compress_streams = 2: FASTQ text -> eBWT, clusters, smoothing -> the ROWS of the edited "
                                      "eBWT (symbols + the replaced rows' originals, qualities) through the codec, no inversion; "
                                      "bfq_stream_ebwt_decode: containers -> LF table -> reads (OUT.fq.dna / OUT.fq.qs)"}
        log(f"ebwt domain: {dt * 1e3:.0f} ms, {nb} bytes, decode {dt2 * 1e3:.0f} ms, equal {same2}")
        eng.fastq_job([text], streams=True, hdr=True, fastq=False, out=zb, compress=3)
        t0 = time.perf_counter()
        z3 = eng.fastq_job([text], streams=True, hdr=True, fastq=False, out=zb, compress=3)
        dt3 = time.perf_counter() - t0
        nb3 = int(len(z3.dna) + len(z3.qs) + len(z3.hdr))
        t0 = time.perf_counter()
        d3, q3, nr3 = eng.ebwt_decode(z3.dna, z3.qs, out=(back[0].array, back[1].array))
        dt4 = time.perf_counter() - t0
        same3 = bool(nr3 == N and np.array_equal(d3, outs["dna"].array[:lens["dna"]]) and np.array_equal(q3, outs["qs"].array[:lens["qs"]]))
        res["ebwt_domain_qs_by_read"] = {"wall_ms": round(dt3 * 1e3, 1), "Mbases_per_s": round(N * L / 1e6 / dt3, 1), "bytes_out": nb3,
                                         "ratio_to_raw_streams": round(tot_raw / max(nb3, 1), 2), "decode_wall_ms": round(dt4 * 1e3, 1),
                                         "decoded_streams_equal_e2e_host": same3,
                                         "what": "compress_streams = 3: bases as rows of the edited eBWT, qualities as the read-order stream (one walk while compressing)"}
        log(f"ebwt domain, qualities by read: {dt3 * 1e3:.0f} ms, {nb3} bytes, decode {dt4 * 1e3:.0f} ms, equal {same3}")
        for b in back:
            b.free()
        for v in zo.values():
            v.free()
    res["total"] = {"raw_bytes": int(tot_raw), "compressed_bytes": int(tot_cmp), "ratio": round(tot_raw / max(tot_cmp, 1), 2),
                    "compress_GB_per_s_host_to_host": round(tot_raw / 1e9 / tot_t, 2)}
    res["what"] = ("OUT.fq.dna / OUT.fq.qs / OUT.h of the e2e_host run -> bfq_stream_compress (read-order DNA: BFQDNAC1, a block-adaptive hashed "
                   "order-K context model + rANS, 1024-base segments; qualities: BFQRANS2, static order-k model + rANS, 8192-symbol segments; names: BFQLINE1 line "
                   "delta + BFQRANS2; one lane per segment) -> bfq_stream_decompress; containers = oracle/bfq_codec_ref.c byte for byte")
    return res


def ebwt_modes(api, eng, text, N, L, log):
    """The reference's tool boundary in process, host arrays (pinned) in and out: step 1 alone (gsufsort / eGap: FASTQ text ->
    eBWT, QS, LCP), then steps 2-4 on that eBWT in bfq_int mode (LCP deduced from the BWT alone, k_bfs.hip) and in bfq_ext
    mode (LCP given, 2 bytes per entry)."""
    n = N * (L + 1)
    pins = [api.PinnedBuffer(n), api.PinnedBuffer(n), api.PinnedBuffer(2 * n), api.PinnedBuffer(N * L), api.PinnedBuffer(N * L), api.PinnedBuffer(8 * (N + 1))]
    bwt, qs = pins[0].array, pins[1].array
    lcp = pins[2].array.view(np.uint16)
    out = (pins[3].array, pins[4].array, pins[5].array.view(np.uint64))
    res = {}
    eng.fastq_build_ebwt(text, out=(bwt, qs, lcp))
    t0 = time.perf_counter()
    eng.fastq_build_ebwt(text, out=(bwt, qs, lcp))
    dt = time.perf_counter() - t0
    res["build_ebwt"] = {"wall_ms": round(dt * 1e3, 1), "Mbases_per_s": round(N * L / 1e6 / dt, 1)}
    log(f"build_ebwt (gsufsort / eGap boundary): {dt * 1e3:.0f} ms")
    # (bfq_ext as BFQzip_ext.py runs it reads eGap's 1-byte LCP file, `--lbytes 1`: entries clamped at 255 -- the same flags for
    # reads below 255 bases, half the upload)
    pin8 = api.PinnedBuffer(n)
    lcp8 = pin8.array[:n]
    lcp8[:] = np.minimum(lcp, 255).astype(np.uint8)
    pins.append(pin8)
    for name, l in (("bfq_int", None), ("bfq_ext", lcp), ("bfq_ext_lbytes1", lcp8)):
        eng.smooth_invert(bwt, qs, l, out=out)                      # warm-up: sizes the workspace
        eng.prof_reset()
        t0 = time.perf_counter()
        _, _, _, st = eng.smooth_invert(bwt, qs, l, out=out)
        dt = time.perf_counter() - t0
        pr = eng.prof()
        res[name] = {"wall_ms": round(dt * 1e3, 1), "Mbases_per_s": round(N * L / 1e6 / dt, 1),
                     "kernel_ms": {k: round(v["ms"], 2) for k, v in sorted(pr.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] >= 0.5},
                     "qs_smoothed": st["qs_smoothed"]}
        log(f"{name} mode: {dt * 1e3:.0f} ms")
    for p in pins:
        p.free()
    res["what"] = ("build_ebwt: bfq_fastq_build_ebwt, FASTQ text -> eBWT + QS + 2-byte LCP (pinned host arrays); bfq_int / bfq_ext: "
                   "bfq_smooth_invert on those arrays -> host reads: upload, [LCP from the BWT | LCP upload], LF table, clusters, "
                   "two LF walks (lengths, reads), download; bfq_ext_lbytes1: the LCP as eGap's 1-byte file holds it (BFQzip_ext.py:172-177)")
    return res


def global_mode(api, parallel, eng, text, N, L, dna_ref, qs_ref, log):
    """parallel.py --global on this one GPU: the collection as ONE eBWT whose two-symbol piles are sorted one
    after the other, position-mode clusters, streams written to /dev/shm -- the per-rank work of the multi-GPU global mode,
    all of it on one rank.  Its streams must equal those of the fused path (e2e_host): parity at full size."""
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        with open(d + "/in.fastq", "wb") as f:
            mv = memoryview(text)
            for o in range(0, len(mv), 1 << 30):
                f.write(mv[o:o + (1 << 30)])
        names = parallel.output_names([d + "/in.fastq"], d + "/G", False)
        runs = []
        for it in range(2):                                      # best of two, as e2e_host is the best of its iterations: a first run
            for nm in names[0].values():                         # pays for device allocations that land on memory the driver has not
                if os.path.exists(nm):                           # cleared yet ("text exchange" 0.03 or 0.5 s at one rank)
                    os.unlink(nm)
            t0 = time.perf_counter()
            tot = parallel.run_global(eng, parallel.Comm(), [d + "/in.fastq"], names, want_fastq=False, want_streams=True)
            runs.append((time.perf_counter() - t0, tot))
        dt, tot = min(runs, key=lambda r: r[0])
        same = bool(np.array_equal(np.fromfile(names[0]["dna"], np.uint8), dna_ref) and np.array_equal(np.fromfile(names[0]["qs"], np.uint8), qs_ref))
        log(f"global mode: {dt:.2f}s (runs: {[round(r[0], 2) for r in runs]}) {tot['seconds']} parity={same}")
        return {"wall_s": round(dt, 3), "runs_s": [round(r[0], 3) for r in runs], "Mbases_per_s": round(N * L / 1e6 / dt, 1), "seconds": tot["seconds"], "streams_equal_fused_path": same,
                "what": "parallel.run_global, 1 rank: file on /dev/shm -> upload, parse, text, 25 piles sorted one by one, position-mode clusters, "
                        "streams written to /dev/shm; compared with e2e_host's streams"}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


_GEN = ("import sys; sys.path.insert(0, %r)\n"
        "from bfqzip_amd import api\n"
        "N, L, dev, path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]\n"
        "e = api.Engine(dev); p = api.PinnedBuffer(N * (2 * L + 30) + 4096)\n"
        "l = e.synth_fastq(api.synth_spec(N, L, seed=20240807), p.array)\n"
        "f = open(path, 'wb'); mv = memoryview(p.array[:l])\n"
        "[f.write(mv[o:o + (1 << 30)]) for o in range(0, l, 1 << 30)]; f.close(); e.close(); p.free(); print(l)\n") % ROOT


def _phases(stderr_text):
    for line in reversed(stderr_text.splitlines()):
        if line.startswith("[bfq phases] "):
            try:
                return json.loads(line[len("[bfq phases] "):])
            except ValueError:
                return None
    return None


def dropin_wall(N, L, params, device, log, keep=False):
    """`gsufsort in.fastq --bwt --qs -o OUT` then `bfq_int -e OUT.bwt -q OUT.bwt.qs -o OUT.fq -m 5` (BFQzip.py:184,215-222) with
    the drop-in executables on /dev/shm files, one right after the other as BFQzip.py runs them: wall seconds per tool
    and each tool's own phase split (its `[bfq phases]` line: exec -> main, lease, HIP start, allocations, file read + H2D,
    GPU, D2H + file write, teardown).  Runs BEFORE this process has created its engine: the input file is written by a
    child process (GPU generator), and the leg starts `settle_s` seconds after that child has gone, so that no tool waits
    for the driver to scrub HBM that the bench itself has just freed."""
    gs = os.path.join(ROOT, "dropin", "external", "gsufsort", "gsufsort")
    bi = os.path.join(ROOT, "dropin", "src_int_mem", "bfq_int")
    if not (os.path.exists(gs) and os.path.exists(bi)):
        return {"skipped": "drop-in executables not built (make -C bfqzip_amd/csrc cli)"}, None
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    ok_keep = False
    try:
        t0 = time.perf_counter()
        tlen = int(subprocess.check_output([sys.executable, "-c", _GEN, str(N), str(L), str(device), d + "/in.fastq"], timeout=600).split()[-1])
        t1 = time.perf_counter()
        settle = 2.0 if N * L > 10**9 else 0.5
        time.sleep(settle)
        env = dict(os.environ, BFQ_M=str(params["M"]), BFQ_B=str(params["B"]), BFQ_TRACE="1", BFQ_DEVICE=str(device))
        ta = time.perf_counter()
        r1 = subprocess.run([gs, d + "/in.fastq", "--bwt", "--qs", "-o", d + "/OUT"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=1200, text=True)
        tb = time.perf_counter()
        if r1.returncode:
            return {"error": "gsufsort: " + r1.stderr[-400:]}, None
        r2 = subprocess.run([bi, "-e", d + "/OUT.bwt", "-q", d + "/OUT.bwt.qs", "-o", d + "/OUT.fq", "-m", str(params["m"])],
                            stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=1200, text=True)
        tc = time.perf_counter()
        if r2.returncode:
            return {"error": "bfq_int: " + r2.stderr[-400:]}, None
        osz = os.path.getsize(d + "/OUT.fq")
        ok = osz == N * (2 * L + 6)                                  # "@" headers: 1 + 1 + L + 1 + 2 + L + 1 per read
        log(f"dropin {N}x{L}: generate {t1 - t0:.2f}s gsufsort {tb - ta:.2f}s bfq_int {tc - tb:.2f}s size_ok={ok}")
        res = {"gsufsort": round(tb - ta, 3), "bfq_int": round(tc - tb, 3), "total": round(tc - ta, 3),
               "Mbases_per_s": round(N * L / 1e6 / (tc - ta), 1), "fastq_bytes": tlen, "output_size_ok": bool(ok),
               "phases": {"gsufsort": _phases(r1.stderr), "bfq_int": _phases(r2.stderr)}, "settle_s": settle,
               "what": "wall seconds of the two processes back to back on /dev/shm files, first GPU work of this bench run "
                       "(input written by a child process that exited settle_s before); phases = each tool's own timeline"}
        ok_keep = keep and ok
        return res, (d if ok_keep else None)
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}, None
    finally:
        if not ok_keep:
            shutil.rmtree(d, ignore_errors=True)


def dropin_parity(d, N, L, dna, qs):
    """OUT.fq of the drop-in leg against the streams of the in-process fused call (e2e_host): lines 2 and 4 of every record."""
    try:
        rec = 2 * L + 6
        fq = np.memmap(d + "/OUT.fq", np.uint8, "r")
        if len(fq) != N * rec:
            return False
        same = True
        step = 1_000_000
        for a in range(0, N, step):
            b = min(N, a + step)
            blk = np.asarray(fq[a * rec:b * rec]).reshape(b - a, rec)
            same = same and np.array_equal(blk[:, 2:2 + L], dna[a * (L + 1):b * (L + 1)].reshape(b - a, L + 1)[:, :L])
            same = same and np.array_equal(blk[:, L + 5:2 * L + 5], qs[a * (L + 1):b * (L + 1)].reshape(b - a, L + 1)[:, :L])
            if not same:
                break
        del fq
        return bool(same)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def device_state(local):
    """Clocks / power cap of the GPU as rocm-smi reports them (why one box runs a pass in 27 ms and another in 33)."""
    if any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) or os.environ.get("ROCPROF_OUTPUT_PATH"):
        return {"skipped": "under rocprofv3 (rocm-smi is a script: an exec from a process whose GPU the profiler has initialised)"}
    try:
        o = subprocess.run(["rocm-smi", "-d", str(local), "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--json"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=30, text=True).stdout
        j = json.loads(o[o.index("{"):])
        card = next(iter(j.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(x in kl for x in ("sclk", "mclk", "fclk", "socclk", "power", "performance level")):
                keep[k] = v
        return keep
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("BFQ_BENCH_WORKLOAD", "30Mx150"))
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("BFQ_BENCH_SCALING", "weak"))
    ap.add_argument("--sample-reads", type=int, default=0, help="reads in the CPU-baseline sample (0: about 40 Mbases)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--piles", type=int, default=0, help="bfq_params.piles: 1 = step 1 pile by pile (13 n bytes of workspace instead of 28.6 n)")
    ap.add_argument("--M", type=int, default=2)
    ap.add_argument("--B", type=int, default=None)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bfqzip_amd import api, fastq, parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("BFQ_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 path on fewer GPUs than ranks
    local = local % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == "nccl" else torch.device("cpu")   # where the few exchanged integers live

    tstart = time.perf_counter()
    Nw, L = parse_workload(args.workload)
    B = args.B if args.B is not None else (1 if (Nw, L) == (30_000_000, 150) else 0)   # BASELINE.json configs[2]: B=1
    par = dict(k=16, m=5, v=ord(">"), f=40, t=20, M=args.M, B=B, piles=args.piles)    # -m 5: what BFQzip.py passes

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - tstart:.1f}s] {msg}", file=sys.stderr, flush=True)

    # the drop-in tools first, before this process holds any HBM (see dropin_wall)
    dropin_res, dropin_dir = {}, None
    if world == 1 and not args.no_dropin:
        dropin_res[f"{Nw}x{L}"], dropin_dir = dropin_wall(Nw, L, par, local, log, keep=not args.no_e2e)
        if (Nw, L) != (1_000_000, 100):
            dropin_res["1000000x100"], _ = dropin_wall(1_000_000, 100, dict(par, B=0), local, log)
    eng = api.Engine(local, **par)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(scaling, steps, warmup):
        """K timed steps in the given mode; returns (seconds, bases this rank processed per step, prof, stats, reads)."""
        if scaling == "strong":
            blocks = parallel.split_blocks(Nw, world)
            mine = [blocks[k] for k in parallel.blocks_of_rank(len(blocks), rank, world)]
            specs = [api.synth_spec(e - s, L, seed=20240807, first=s, collection=Nw) for s, e in mine]
        else:
            specs = [api.synth_spec(Nw, L, seed=20240807 + rank)]
        bufs = []
        for sp in specs:
            n = int(sp.N); tot = n * L
            db = torch.empty(tot, dtype=torch.uint8, device=dev); dq = torch.empty_like(db)
            dr = torch.empty(n + 1, dtype=torch.int64, device=dev)
            ob = torch.empty_like(db); oq = torch.empty_like(db)
            eng.synth_device(sp, db.data_ptr(), dq.data_ptr(), dr.data_ptr())
            bufs.append((n, tot, db, dq, dr, ob, oq))
        torch.cuda.synchronize()

        def step():
            st = None
            for n, tot, db, dq, dr, ob, oq in bufs:
                st = eng.run_reads_device(db.data_ptr(), dq.data_ptr(), dr.data_ptr(), n, tot, ob.data_ptr(), oq.data_ptr())
            return st
        st = None
        for i in range(warmup):
            st = step()
            log(f"[{scaling}] warmup step {i} done, workspace {eng.workspace_bytes() / 2**30:.1f} GiB")
        eng.prof_trace_select("k_radix_scatter")
        eng.prof_reset()
        step_ms = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts = time.perf_counter()
            st = step()
            step_ms.append((time.perf_counter() - ts) * 1e3)      # the call returns synchronised (it reads the counters back)
            if world > 1:                                   # the only exchange: per-block output sizes (one integer per rank)
                sz = torch.tensor([sum(b[1] for b in bufs)], dtype=torch.int64, device=cdev)
                lst = [torch.empty_like(sz) for _ in range(world)]
                dist.all_gather(lst, sz)
        barrier()
        dt = time.perf_counter() - t0
        bases = sum(b[1] for b in bufs)
        if world > 1:
            t = torch.tensor([dt, float(bases)], dtype=torch.float64, device=cdev)
            tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ts = t.clone(); dist.all_reduce(ts, op=dist.ReduceOp.SUM)
            dt, bases_all = float(tm[0].item()), float(ts[1].item())
        else:
            bases_all = float(bases)
        prof = eng.prof()
        prof["_step_ms"] = step_ms
        prof["_scatter_ms"] = [float(x) for x in eng.prof_trace()]
        eng.prof_trace_select(None)
        reads = sum(b[0] for b in bufs)
        del bufs
        torch.cuda.empty_cache()
        return dt, bases_all, prof, st, reads

    dt, bases_all, prof, st, reads_rank = measure(args.scaling, args.steps, args.warmup)
    step_ms, scatter_ms = prof.pop("_step_ms"), prof.pop("_scatter_ms")
    log(f"{args.steps} timed steps ({args.scaling}): {dt:.3f}s")
    strong_extra = None
    if world > 1 and args.scaling == "weak":                # configs[3] beside the weak figure
        sdt, sbases, _, _, sreads = measure("strong", args.steps, max(1, args.warmup))
        strong_extra = {"value": round(sbases / 1e6 / (sdt / args.steps), 2), "unit": "Mbases/s", "ms_per_step": round(sdt / args.steps * 1e3, 3),
                        "reads_total": Nw, "reads_rank0": sreads, "what": "BASELINE configs[3]: one collection cut by split_blocks(reads, N), one block per rank"}
        log(f"strong (configs[3]) {args.steps} steps: {sdt:.3f}s")

    rc = 0
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = bases_all / 1e6 / (dt / args.steps)
        rows_rank = reads_rank * (L + 1)
        # dominant kernel = largest accumulated HIP-event time over the timed region
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        dname, d = dom
        avg_ms = d["ms"] / d["launches"]
        ach = d["alg_bytes"] / d["launches"] / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dname, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dname, rows_rank) if (Nw, L) == (30_000_000, 150) and reads_rank == Nw else None,
                "avg_launch_ms": round(avg_ms, 4), "launches": int(d["launches"]),
                "alg_bytes_per_launch": d["alg_bytes"] / d["launches"],
                "job_alg_bytes_per_base": round(alg_bytes_per_base(L), 1),
                "job_frac": round(alg_bytes_per_base(L) * (bases_all / world / (dt / args.steps)) / (HBM_PEAK_GBS * 1e9), 4)}
        # kernels within 10 % of the dominant one's accumulated time (the five scatter passes and the one refinement launch
        # trade places from run to run): named, so that a change of `roofline.kernel` between two lines is not read as a change of code
        roof["co_dominant"] = {k: {"ms_per_step": round(v["ms"] / args.steps, 3), "launches_per_step": v["launches"] / args.steps,
                                   "frac": round(v["alg_bytes"] / v["launches"] / (v["ms"] / v["launches"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                               for k, v in prof.items() if v["ms"] >= 0.9 * d["ms"] and v["alg_bytes"] > 0}
        kern = {k: round(v["ms"] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        # the same figure for the other heavy kernels (algorithmic GB/s and fraction of the HBM peak)
        # the sort passes one by one (launch order: KEY_PASSES scatters per step; pass 0 reads the records k_build_keys wrote)
        if scatter_ms and len(scatter_ms) % args.steps == 0:
            pp = len(scatter_ms) // args.steps
            a = np.asarray(scatter_ms, np.float64).reshape(args.steps, pp)
            roof["scatter_passes_ms"] = {"per_step": pp, "median": [round(float(x), 3) for x in np.median(a, axis=0)],
                                         "min": [round(float(x), 3) for x in a.min(axis=0)], "max": [round(float(x), 3) for x in a.max(axis=0)]}
        # Three ways to price the whole job against the 8 TB/s peak, side by side:
        #   job_frac       SURVEY 8(d)'s A(L) (an 8-pass sort of 16-byte records, a 64-byte rank block per LF step)
        #   job_frac_impl  the bytes THIS implementation must move: the kernels' own algorithmic bytes (5 passes of 12-byte
        #                  records ...), the LF walk priced at the 8-byte table entry + 2 bytes written per base it really needs
        #   job_frac_pmc   the HBM traffic rocprofv3's FETCH_SIZE / WRITE_SIZE counters measured (profiles/*/traffic_per_row.json)
        inv = prof.get("k_invert", {"alg_bytes": 0.0})
        bases_rank = reads_rank * L
        impl_bytes = (sum(v["alg_bytes"] for v in prof.values()) - inv["alg_bytes"]) / args.steps + 10.0 * bases_rank
        rate_rows = rows_rank / (dt / args.steps)
        roof["impl_alg_bytes_per_row"] = round(impl_bytes / max(rows_rank, 1), 1)
        roof["job_frac_impl"] = round(impl_bytes / max(rows_rank, 1) * rate_rows / (HBM_PEAK_GBS * 1e9), 4)
        tpr = pmc_traffic_per_row()
        if tpr and (Nw, L) == (30_000_000, 150):
            roof["pmc_bytes_per_row"] = round(tpr, 1)
            roof["job_frac_pmc"] = round(tpr * rate_rows / (HBM_PEAK_GBS * 1e9), 4)
        roof["k_invert_need"] = {"alg_bytes_per_base_survey": 68.0, "alg_bytes_per_base_impl": 10.0,
                                 "frac_impl": round(10.0 * bases_rank / (inv["ms"] / max(inv.get("launches", 1), 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if inv.get("ms") else None,
                                 "what": "SURVEY prices an LF step at a 64-byte rank block; the tabulated LF entry is 8 bytes (+ 2 written): a random 8-byte read still fetches a 64-byte sector"}
        roof["by_kernel"] = {k: {"avg_launch_ms": round(v["ms"] / v["launches"], 3),
                                 "achieved": round(v["alg_bytes"] / v["launches"] / (v["ms"] / v["launches"] * 1e-3) / 1e9, 1),
                                 "frac": round(v["alg_bytes"] / v["launches"] / (v["ms"] / v["launches"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                             for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:5] if v["ms"] > 0 and v["alg_bytes"] > 0}
        per = "per GPU" if args.scaling == "weak" else f"in total, cut into {world} block(s) by the BFQzip_parallel split"
        res = {"metric": "Mbases/s end-to-end (eBWT+cluster+LF), reads and results resident in HBM", "value": round(value, 2), "unit": "Mbases/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u64",
               "data": "synthetic (seeded generator, 30x coverage, 1% errors, 0.1% N)",
               "config": {"workload": f"{Nw}x{L}bp synthetic reads {per}, M={args.M} B={B} -m 5 K=16", "reads_rank0": reads_rank,
                          "read_len": L, "rows_rank0": rows_rank, "parallelism": f"{world} independent blocks"},
               "roofline": roof, "kernel_ms_per_step": kern,
               "stats": {k: st[k] for k in ("num_clust", "bases_inside", "qs_smoothed", "modified", "n_segments", "n_big_segments")},
               "workspace_gib": round(eng.workspace_bytes() / 2**30, 2),
               "step_ms": {"min": round(min(step_ms), 3), "median": round(float(np.median(step_ms)), 3), "max": round(max(step_ms), 3)},
               "device_state": device_state(local)}
        if strong_extra:
            res["strong"] = strong_extra
        if world == 1:
            text = None
            if not args.no_e2e:
                try:
                    sp = api.synth_spec(Nw, L, seed=20240807)
                    res["e2e_host"], pin, tlen, outs = e2e_host(api, eng, sp, Nw, L, 3, log)
                    text = pin.array[:tlen]
                    if dropin_dir is not None:                  # parity of the drop-in leg's OUT.fq, checked while the streams exist
                        dropin_res[f"{Nw}x{L}"]["fastq_equals_fused_streams"] = dropin_parity(dropin_dir, Nw, L, outs["dna"].array, outs["qs"].array)
                        dropin_dir = None
                except Exception as e:                       # e.g. pinned memory refused: reported, not fatal
                    res["e2e_host"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.no_e2e and text is not None and "error" not in res.get("e2e_host", {}):
                res["global_mode"] = global_mode(api, parallel, eng, text, Nw, L, outs["dna"].array[:Nw * (L + 1)], outs["qs"].array[:Nw * (L + 1)], log)
            if not args.no_e2e and text is not None and "error" not in res.get("e2e_host", {}):
                try:
                    nl = Nw * (L + 1)
                    lens = {"dna": nl, "qs": nl, "hdr": res["e2e_host"]["bytes_out"] - 2 * nl}
                    res["stream_codec"] = stream_codec(api, eng, outs, lens, Nw, L, log, not args.no_cpu, text)
                except Exception as e:
                    res["stream_codec"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.no_e2e:
                try:
                    for v in outs.values():
                        v.free()
                    res["ebwt_modes"] = ebwt_modes(api, eng, text, Nw, L, log)
                except Exception as e:
                    res["ebwt_modes"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.no_cpu:
                from oracle import orc
                cb, (sb, sq, sr, sout) = cpu_baseline(api, orc, L, 20240807, min(args.sample_reads or max(1000, 40_000_000 // L), Nw), par)
                res["cpu_baseline"] = cb
                # the GPU path on the same sample must reproduce the CPU output byte for byte
                gb, gq, gst = eng.run_reads(sb, sq, sr)
                res["sample_parity"] = bool(fastq.format_fastq(gb, gq, sr) == sout)
                if not res["sample_parity"]:
                    rc = 3
                    res["value"] = None                      # a number whose output differs from the reference's is not a result
            if dropin_dir is not None:
                shutil.rmtree(dropin_dir, ignore_errors=True)
            if dropin_res:
                res["dropin_wall_s"] = dropin_res
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
