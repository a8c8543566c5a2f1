#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference objects (oracle/_ref/ref_driver).

Run in the build container only (needs /root/reference to have built oracle/_ref).  Inputs are
regenerated from seeds by scalce_amd.synth, so the fixtures hold expected OUTPUTS plus the
parameters; the FASTQ text itself is stored only for the tiny hand-made cases.
What each array is: see tests/golden/README.md.
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scalce_amd import synth  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

CASES = {
    # name: (n, L, seed, synth kwargs, pattern text or None, lossy LUT?, keep_full)
    "se100": dict(n=3000, L=100, seed=11, kw=dict(dup_frac=0.2, n_frac=0.01), ptxt=None, lossy=0),
    "se100_lossy": dict(n=3000, L=100, seed=11, kw=dict(dup_frac=0.2, n_frac=0.01), ptxt=None, lossy=30),
    "se150_text": dict(n=1500, L=150, seed=12, kw=dict(dup_frac=0.1), ptxt="mixed", lossy=0),
    "se36_ties": dict(n=4000, L=36, seed=13, kw=dict(dup_frac=0.3), ptxt="fourmers", lossy=0),
    "se100_110k": dict(n=110000, L=100, seed=14, kw=dict(), ptxt=None, lossy=0, hash_only=True),
    # shrink factor of compress.cpp:297-313 (every GPU config of BASELINE.json has factor >= 2: C2 2, C3 7, C4 24)
    "se100_f7": dict(n=3000, L=100, seed=11, kw=dict(dup_frac=0.2, n_frac=0.01), ptxt=None, lossy=0, factor=7),
    # paired (-r): mate 2 through the reference's output_read(.., 0, 0) / output_quality(.., ZZ = 1)
    "pe150": dict(n=2500, L=150, seed=15, seed2=16, kw=dict(dup_frac=0.1, n_frac=0.005), ptxt=None, lossy=0),
    "pe150_lossy_f2": dict(n=2500, L=150, seed=17, seed2=18, kw=dict(dup_frac=0.1, n_frac=0.005), ptxt=None, lossy=30, factor=2),
}


def pattern_text(kind):
    import random
    rnd = random.Random(4242)
    if kind == "fourmers":  # every 4-mer: each position is an equal-level tie
        return "\n".join("".join(x) for x in __import__("itertools").product("ACGT", repeat=4)) + "\n"
    pats = []
    for ln, cnt in ((6, 40), (9, 300), (14, 800)):
        for _ in range(cnt):
            pats.append("".join(rnd.choice("ACGT") for _ in range(ln)))
    # nested cores (a core that is a prefix / suffix of another) and one duplicate
    pats += [pats[400][:7], pats[401][3:], pats[5], "ACGTACGTACGT", "ACGTACGT"]
    return "\n".join(pats) + "\n"


def run_case(name, c):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    bases, quals = synth.reads_and_quals(c["n"], c["L"], seed=c["seed"], **c["kw"])
    paired = "seed2" in c
    if paired:
        bases2, quals2 = synth.reads_and_quals(c["n"], c["L"], seed=c["seed2"], **c["kw"])
    with tempfile.TemporaryDirectory() as d:
        fq = os.path.join(d, "in_1.fq")
        if paired:
            open(fq, "wb").write(synth.fastq_bytes_fast(bases, quals, prefix="p.", suffix="/1"))
            fq2 = os.path.join(d, "in_2.fq")
            open(fq2, "wb").write(synth.fastq_bytes_fast(bases2, quals2, prefix="p.", suffix="/2"))
        else:
            open(fq, "wb").write(synth.fastq_bytes_fast(bases, quals))
        args = [DRIVER, fq, d]
        if paired:
            args += ["-2", fq2]
        if c.get("factor", 1) != 1:
            args += ["-f", str(c["factor"])]
        ptxt = None
        if c["ptxt"]:
            ptxt = pattern_text(c["ptxt"])
            open(os.path.join(d, "p.txt"), "w").write(ptxt)
            args += ["-P", os.path.join(d, "p.txt")]
        lut = None
        if c["lossy"]:
            stat = np.bincount(quals.reshape(-1), minlength=128).astype(np.int32)
            off, vals = O.qmap_init(stat, c["lossy"])
            lut = np.concatenate([[off], vals]).astype(np.int32)
            open(os.path.join(d, "q.txt"), "w").write(" ".join(map(str, lut)))
            args += ["-q", os.path.join(d, "q.txt")]
            if paired:  # the model of mate 2 comes from mate 2's own sample (get_quality_stats, compress.cpp:554-584)
                stat2 = np.bincount(quals2.reshape(-1), minlength=128).astype(np.int32)
                off2, vals2 = O.qmap_init(stat2, c["lossy"])
                lut2 = np.concatenate([[off2], vals2]).astype(np.int32)
                open(os.path.join(d, "q2.txt"), "w").write(" ".join(map(str, lut2)))
                args += ["-q2", os.path.join(d, "q2.txt")]
        subprocess.run(args, check=True)
        tok = np.fromfile(os.path.join(d, "tok.i32"), dtype=np.int32).reshape(-1, 2)
        order = np.fromfile(os.path.join(d, "order.i64"), dtype=np.int64)
        ids = np.fromfile(os.path.join(d, "ids.i32"), dtype=np.int32)
        packed = np.fromfile(os.path.join(d, "packed.bin"), dtype=np.uint8)
        names = np.fromfile(os.path.join(d, "names.bin"), dtype=np.uint8)
        qual = np.fromfile(os.path.join(d, "qual.bin"), dtype=np.uint8)
        freq4 = np.fromfile(os.path.join(d, "freq4.u64"), dtype=np.uint64)
        ac = np.fromfile(os.path.join(d, "ac.bin"), dtype=np.uint8)
        table = np.fromfile(os.path.join(d, "table.u32"), dtype=np.uint32)
        if paired:
            packed2 = np.fromfile(os.path.join(d, "packed2.bin"), dtype=np.uint8)
            qual2 = np.fromfile(os.path.join(d, "qual2.bin"), dtype=np.uint8)
            freq4_2 = np.fromfile(os.path.join(d, "freq4_2.u64"), dtype=np.uint64)
            ac2 = np.fromfile(os.path.join(d, "ac2.bin"), dtype=np.uint8)
            table2 = np.fromfile(os.path.join(d, "table2.u32"), dtype=np.uint32)
    sha = lambda a: hashlib.sha256(a.tobytes()).hexdigest()
    out = dict(n=c["n"], L=c["L"], seed=c["seed"], kw=repr(c["kw"]), lossy=c["lossy"],
               sha_tok=sha(tok), sha_order=sha(order), sha_packed=sha(packed), sha_names=sha(names),
               sha_qual=sha(qual), sha_freq4=sha(freq4), sha_ac=sha(ac), ac_len=len(ac),
               factor=c.get("factor", 1), sha_table=sha(table))
    if paired:
        out.update(seed2=c["seed2"], sha_packed2=sha(packed2), sha_qual2=sha(qual2), sha_freq4_2=sha(freq4_2),
                   sha_ac2=sha(ac2), ac2_len=len(ac2), sha_table2=sha(table2), ac2_head=ac2[:4096])
        if lut is not None:
            out["lut2"] = lut2
    if ptxt is not None:
        out["ptxt"] = ptxt
    if lut is not None:
        out["lut"] = lut
    if not c.get("hash_only"):
        nz = np.flatnonzero(freq4 != 1)
        out.update(tok=tok, order=order.astype(np.int32), ids=ids, packed=packed,
                   freq4_idx=nz.astype(np.int32), freq4_val=freq4[nz].astype(np.int64), ac_head=ac[:4096])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("sha") is False and k not in ("ptxt",)})


if __name__ == "__main__":
    if not os.path.exists(DRIVER):
        sys.exit("oracle/_ref/ref_driver missing: run `make -C oracle` where /root/reference exists")
    for name, c in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        run_case(name, c)
