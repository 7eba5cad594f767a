#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REFERENCE itself (run in the build container only).

Inputs : example/reads.fastq of the reference (a data file) and two seeded
         synthetic sets (bfq_synth_host of libbfqhip.so, host code only).
eBWT   : built by the oracle's suffix sorter (gsufsort is an absent submodule);
         pinned by the round trip `reference bfq_int -k 10000` == input reads.
Outputs: FASTQ written by the reference bfq_int compiled by oracle/Makefile into
         oracle/_ref/ (one binary per -DM/-DB), for every (M,B) and a few flag sets.
Only data is stored: inputs, eBWT/QS/LCP bytes, expected FASTQ (or its md5).
"""
import hashlib, json, os, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from bfqzip_amd import api, fastq
from oracle import orc

REF = "/root/reference"


def run_ref(M, B, bwt, qs, flags, headers=None):
    with tempfile.TemporaryDirectory() as d:
        open(d + "/x.bwt", "wb").write(bwt.tobytes()); open(d + "/x.bwt.qs", "wb").write(qs.tobytes())
        cmd = [orc.ref_binary(M, B), "-e", d + "/x.bwt", "-q", d + "/x.bwt.qs", "-o", d + "/o.fq"] + flags
        if headers is not None:
            open(d + "/x.h", "wb").write(b"".join(h + b"\n" for h in headers))
            cmd += ["-H", d + "/x.h"]
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, timeout=120)
        return open(d + "/o.fq", "rb").read()


def main():
    orc.build()
    sets = {}
    b, q, r, h = fastq.read_fastq(REF + "/example/reads.fastq")
    sets["example"] = (b, q, r, h)
    b1, q1, r1, h1 = fastq.read_fastq(REF + "/example/reads_1.fastq")
    b2, q2, r2, h2 = fastq.read_fastq(REF + "/example/reads_2.fastq")
    sets["paired"] = (np.concatenate([b1, b2]), np.concatenate([q1, q2]),
                      np.concatenate([r1, r2[1:] + r1[-1]]), h1 + h2)   # BFQzip_parallel.py:325-360 appends mate block
    sp = api.synth_spec(2000, 30, Lmax=60, seed=11, coverage=30, err_ppm=20000, n_ppm=15000, snp_every=97, dsnp_every=131)
    sets["synth_var"] = api.synth_host(sp) + (None,)
    sp = api.synth_spec(1500, 60, seed=12, coverage=30)
    sets["synth_fix"] = api.synth_host(sp) + (None,)
    index = {}
    for name, (b, q, r, h) in sets.items():
        open(f"{HERE}/{name}.fastq", "wb").write(fastq.format_fastq(b, q, r, h))
        bwt, qs, lcp = orc.build_ebwt(b, q, r)
        open(f"{HERE}/{name}.bwt", "wb").write(bwt.tobytes())
        open(f"{HERE}/{name}.bwt.qs", "wb").write(qs.tobytes())
        open(f"{HERE}/{name}.lcp16", "wb").write(lcp.astype(np.uint16).tobytes())
        ent = {"n": int(len(bwt)), "reads": int(len(r) - 1), "bwt_md5": hashlib.md5(bwt.tobytes()).hexdigest(),
               "qs_md5": hashlib.md5(qs.tobytes()).hexdigest(), "out": {}}
        # identity: the reference inverts our eBWT back to the input (pins the step-1 contract)
        ident = run_ref(2, 0, bwt, qs, ["-k", "10000"])
        assert ident == fastq.format_fastq(b, q, r, None), name
        cases = [(M, B, ["-m", "5"]) for M in range(4) for B in range(2)]
        cases += [(2, 0, ["-m", "2"]), (2, 0, ["-m", "5", "-k", "8", "-v", "53", "-t", "35", "-f", "50"]),
                  (1, 1, ["-m", "3", "-k", "3", "-t", "5"]), (0, 0, ["-m", "9", "-k", "5", "-f", "70", "-v", "73"])]
        for M, B, flags in cases:
            out = run_ref(M, B, bwt, qs, flags)
            key = f"M{M}B{B} " + " ".join(flags)
            ent["out"][key] = hashlib.md5(out).hexdigest()
            if (M, B) == (2, 0) and flags == ["-m", "5"]:
                open(f"{HERE}/{name}.M2B0.fq", "wb").write(out)
        if h is not None:
            out = run_ref(2, 0, bwt, qs, ["-m", "5"], headers=h)
            ent["out"]["M2B0 -m 5 -H"] = hashlib.md5(out).hexdigest()
        index[name] = ent
        print(name, ent["n"], len(ent["out"]))
    json.dump(index, open(f"{HERE}/index.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
