"""Whole-file cases shared by tests/golden/make_golden_files.py and tests/test_ref_files.py.

One flag list per case drives three tools: the reference itself (oracle/_ref/ref_full: the reference's own
compress() / decompress(), /root/reference/compress.cpp:721, decompress.cpp:79), the CPU restatement (oracle/orc_cli)
and the product (scalce_amd/bin/scalce).  Inputs are regenerated from seeds, so the fixture (tests/golden/files.json)
holds only hashes of what the reference wrote.
"""
import gzip
import hashlib
import os
import subprocess

from scalce_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FULL = os.path.join(ROOT, "oracle", "_ref", "ref_full")
ORC_CLI = os.path.join(ROOT, "oracle", "orc_cli")
SCALCE = os.path.join(ROOT, "scalce_amd", "bin", "scalce")
PBIN = os.path.join(ROOT, "tests", "golden", "patterns.bin")

SE = dict(n=30000, L=100, seed=21, kw=dict(dup_frac=0.2, n_frac=0.01))
PE = dict(n=8000, L=150, seed=22, seed2=23, kw=dict(dup_frac=0.1, n_frac=0.005))

# flags in orc_cli / ref_full spelling (-B in bytes); `files` = record counts of several input files of one run
CASES = {
    "se100": dict(SE, flags=[]),
    "se100_p30": dict(SE, flags=["-p", "30"]),
    "se100_B": dict(SE, flags=["-B", "1048576"]),            # 6 spill chunks (compress.cpp:708-715) and their merge
    "se100_A": dict(SE, flags=["-A"]),
    "se100_nlib": dict(SE, flags=["-n", "lib"]),
    "se100_s1000": dict(SE, flags=["-s", "1000", "-p", "30"]),  # lossy map from the first 1000 records only
    "se100_ptxt": dict(SE, flags=[], ptxt="mixed"),          # -P text core list (reads.cpp:379-410)
    "se100_gz": dict(SE, flags=["-c", "gz"]),
    "se100_gzin": dict(SE, flags=[], gz_input=True),         # gzipped FASTQ in (compress.cpp:756: every input goes through gz)
    "pe150": dict(PE, flags=["-r"]),
    "pe150_p30_B": dict(PE, flags=["-r", "-p", "30", "-B", "1048576"]),
    "pe150_gz_nlib": dict(PE, flags=["-r", "-c", "gz", "-n", "run7"]),
    # several input files in one run; the first holds fewer records than -s: the quality sample stops at its end
    # (get_quality_stats reads files[0] only, compress.cpp:761, qualities.cpp:66-78)
    "multi": dict(SE, flags=["-s", "5000", "-p", "30"], files=[700, 19300, 10000], oracle=False),
    # what real FASTQ holds beside ACGTN: lower-case bases (soft-masked) and IUPAC ambiguity letters -- getval (const.cpp:47-49,
    # const.h:127) maps a c g t like A C G T and every other letter to 0 (= A); only an upper-case 'N' zeroes the quality
    # (qualities.cpp:183), and a base whose quality maps to 0 comes back as 'N' (decompress.cpp:350-351)
    "se100_letters": dict(SE, flags=[], letters=True),
    "se100_letters_p30": dict(SE, flags=["-p", "30"], letters=True),
    # Phred+64 qualities (Illumina 1.3-1.7): no character below 64 in the sample -> offset 64 (qualities.cpp:99-104)
    "se100_phred64": dict(SE, flags=[], phred64=True),
    "se100_phred64_p30": dict(SE, flags=["-p", "30"], phred64=True),
}


def pattern_text(kind):
    import random
    rnd = random.Random(4242)
    assert kind == "mixed"
    pats = []
    for ln, cnt in ((6, 40), (9, 300), (14, 800)):
        for _ in range(cnt):
            pats.append("".join(rnd.choice("ACGT") for _ in range(ln)))
    pats += [pats[400][:7], pats[401][3:], pats[5], "ACGTACGTACGT", "ACGTACGT"]
    return "\n".join(pats) + "\n"


def paired(name):
    return "-r" in CASES[name]["flags"]


def input_names(name):
    c = CASES[name]
    ext = ".fq.gz" if c.get("gz_input") else ".fq"
    if "files" in c:
        return ["in%c_1%s" % (97 + k, ext) for k in range(len(c["files"]))]
    return ["in_1" + ext]


def write_inputs(name, d):
    """FASTQ text of the case into directory d; returns the concatenated text per mate (what a decoder must restore,
    names and '+' lines being canonical in the generator)."""
    c = CASES[name]
    d = str(d)
    mates = []
    for m, seed in enumerate([c["seed"]] + ([c["seed2"]] if paired(name) else [])):
        bases, quals = synth.reads_and_quals(c["n"], c["L"], seed=seed, **c["kw"])
        if c.get("letters"):
            import numpy as np
            rng = np.random.default_rng(seed + 1000)
            low = rng.random(bases.shape) < 0.12                     # soft-masked stretches and single bases, 'n' among them
            low[rng.integers(0, c["n"], 200), :] = True              # (whole reads in lower case as well)
            bases = np.where(low, bases | 0x20, bases)
            amb = np.frombuffer(b"RYKMSWBDHVUXryksw", dtype=np.uint8)
            hit = rng.random(bases.shape) < 0.01
            bases = np.where(hit, amb[rng.integers(0, len(amb), bases.shape)], bases)
        if c.get("phred64"):
            quals = quals + 31
        if paired(name):
            text = synth.fastq_bytes_fast(bases, quals, prefix="p.", suffix="/%d" % (m + 1))
        else:
            text = synth.fastq_bytes_fast(bases, quals)
        mates.append(text)
        if "files" in c:
            lines = text.split(b"\n")
            at = 0
            for k, cnt in enumerate(c["files"]):
                piece = b"\n".join(lines[4 * at:4 * (at + cnt)]) + b"\n"
                open(os.path.join(d, input_names(name)[k]), "wb").write(piece)
                at += cnt
            assert at == c["n"]
        else:
            fn = os.path.join(d, input_names(name)[0].replace("_1", "_%d" % (m + 1)))
            data = gzip.compress(text, 1) if c.get("gz_input") else text
            open(fn, "wb").write(data)
    if c.get("ptxt"):
        open(os.path.join(d, "p.txt"), "w").write(pattern_text(c["ptxt"]))
    return mates


def run_tool(tool, name, d, prefix, check=True):
    """compress the case's inputs with `tool` ('ref' | 'orc' | 'hip') into d/<prefix>_<mate>.scalce{n,r,q}"""
    c = CASES[name]
    d = str(d)
    ins = [os.path.join(d, f) for f in input_names(name)]
    out = os.path.join(d, prefix)
    table = ["-P", os.path.join(d, "p.txt")] if c.get("ptxt") else [PBIN]
    flags = list(c["flags"])
    if tool == "ref":
        cmd = [REF_FULL, "compress", *table, ",".join(ins), out, *flags, "-T", "1", "-t", os.path.join(d, "tmp_" + prefix)]
    elif tool == "orc":
        assert len(ins) == 1
        cmd = [ORC_CLI, "compress", *table, ins[0], out, *flags]
    else:
        if "-B" in flags:
            i = flags.index("-B")
            assert int(flags[i + 1]) % (1 << 20) == 0
            flags[i + 1] = "%dM" % (int(flags[i + 1]) >> 20)
        if "-c" not in flags:
            flags += ["-c", "no"]
        table = table if c.get("ptxt") else ["--patterns-bin", PBIN]
        cmd = [SCALCE, *flags, "-o", out, *ins, *table]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if check:
        assert r.returncode == 0, (cmd, r.stderr[-2000:])
    return r


def run_decompress(tool, name, d, prefix, out_prefix, check=True):
    """decompress d/<prefix>_1.scalcen with `tool` into d/<out_prefix>_<mate>.fastq"""
    c = CASES[name]
    d = str(d)
    src = os.path.join(d, prefix + "_1.scalcen")
    out = os.path.join(d, out_prefix)
    table = ["-P", os.path.join(d, "p.txt")] if c.get("ptxt") else [PBIN]
    dflags = ["-r"] if paired(name) else []
    if tool == "ref":
        cmd = [REF_FULL, "decompress", *table, src, out, *dflags, "-T", "1"]
    elif tool == "orc":
        cmd = [ORC_CLI, "decompress", *table, src, out, *dflags]
    else:
        table = table if c.get("ptxt") else ["--patterns-bin", PBIN]
        cmd = [SCALCE, "-d", *dflags, "-o", out, src, *table]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if check:
        assert r.returncode == 0, (cmd, r.stderr[-2000:])
    return r


def content(path):
    raw = open(path, "rb").read()
    return gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw


def archive_hashes(name, d, prefix):
    out = {}
    for m in ((1, 2) if paired(name) else (1,)):
        for ext in "nrq":
            out["%d.scalce%s" % (m, ext)] = hashlib.sha256(content(os.path.join(str(d), "%s_%d.scalce%s" % (prefix, m, ext)))).hexdigest()
    return out


def fastq_hashes(name, d, out_prefix):
    return {"%d.fastq" % m: hashlib.sha256(open(os.path.join(str(d), "%s_%d.fastq" % (out_prefix, m)), "rb").read()).hexdigest()
            for m in ((1, 2) if paired(name) else (1,))}


def hash_outputs(name, d, prefix):
    """archives written by run_tool(.., prefix) + the FASTQ the REFERENCE restores from them"""
    h = archive_hashes(name, d, prefix)
    run_decompress("ref", name, d, prefix, prefix + "_back")
    h.update(fastq_hashes(name, d, prefix + "_back"))
    return h
