"""-m gpu: the `scalce` command line (C++ host over the C ABI) against the oracle's files, both directions."""
import gzip
import os
import sys
import subprocess

import pytest

import oraclelib as O
from scalce_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "scalce_amd", "bin", "scalce")
PBIN = os.path.join(ROOT, "tests", "golden", "patterns.bin")


def run_cli(*args):
    r = subprocess.run([CLI, *map(str, args)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


def maybe_gunzip(path):
    raw = open(path, "rb").read()
    return gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw


@pytest.mark.parametrize("flags", [[], ["-r"], ["-p", "30"], ["-A"], ["-n", "lib"], ["-c", "gz"], ["-B", "1M"],
                                   ["-r", "-c", "gz", "-p", "10"]])
def test_cli_compress_matches_oracle_and_decompress_roundtrips(flags, tmp_path):
    paired = "-r" in flags
    n, L = 8000, 100
    synth.write_fastq(str(tmp_path / "in_1.fq"), n, L, seed=31, n_frac=0.003, dup_frac=0.1,
                      paired_suffix="/1" if paired else None)
    if paired:
        synth.write_fastq(str(tmp_path / "in_2.fq"), n, L, seed=32, paired_suffix="/2")
    oflags = [("1048576" if f == "1M" else f) for f in flags]
    cflags = list(flags) if "-c" in flags else list(flags) + ["-c", "no"]
    run_cli(*cflags, "-o", tmp_path / "hip", tmp_path / "in_1.fq", "--patterns-bin", PBIN)
    O.orc_cli("compress", PBIN, tmp_path / "in_1.fq", tmp_path / "orc", *oflags)
    for m in ((1, 2) if paired else (1,)):
        for ext in "nrq":
            a = maybe_gunzip(tmp_path / f"orc_{m}.scalce{ext}")
            h = maybe_gunzip(tmp_path / f"hip_{m}.scalce{ext}")
            assert a == h, f"{flags} .scalce{ext} mate {m}: {len(h)} vs {len(a)} bytes"
    dflags = (["-r"] if paired else []) + (["-n", "lib"] if "-n" in flags else [])
    run_cli("-d", *dflags, "-o", tmp_path / "back", tmp_path / "hip_1.scalcen", "--patterns-bin", PBIN)
    O.orc_cli("decompress", PBIN, tmp_path / "orc_1.scalcen", tmp_path / "oback", *dflags)
    for m in ((1, 2) if paired else (1,)):
        assert open(tmp_path / f"back_{m}.fastq", "rb").read() == open(tmp_path / f"oback_{m}.fastq", "rb").read()


def test_cli_errors_like_the_reference(tmp_path):
    synth.write_fastq(str(tmp_path / "in_1.fq"), 100, 50, seed=1)
    r = subprocess.run([CLI, str(tmp_path / "in_1.fq")], capture_output=True, text=True)
    assert r.returncode == 1 and "(ERROR) No output file specified." in r.stderr
    r = subprocess.run([CLI, "-o", "x", str(tmp_path / "nope_1.fq")], capture_output=True, text=True)
    assert r.returncode == 1 and "does not exist" in r.stderr
    lines = open(tmp_path / "in_1.fq", "rb").read().split(b"\n")
    lines[9] = lines[9][:-3]
    open(tmp_path / "bad_1.fq", "wb").write(b"\n".join(lines))
    r = subprocess.run([CLI, "-o", str(tmp_path / "o"), str(tmp_path / "bad_1.fq"), "--patterns-bin", PBIN],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "(ERROR)" in r.stderr


def test_cli_split_output_and_device_records(tmp_path):
    """-S: the device-built text is cut at record boundaries (decompress.cpp:276-287); and the C entry behind the CLI's
    decompressor, scalce_fastq_records, rebuilds the oracle's FASTQ from the archive streams when called directly."""
    import numpy as np
    import struct
    import torch
    from scalce_amd import host
    n, L = 5000, 100
    synth.write_fastq(str(tmp_path / "in_1.fq"), n, L, seed=41, n_frac=0.004, dup_frac=0.1)
    run_cli("-c", "no", "-o", tmp_path / "a", tmp_path / "in_1.fq", "--patterns-bin", PBIN)
    run_cli("-d", "-o", tmp_path / "whole", tmp_path / "a_1.scalcen", "--patterns-bin", PBIN)
    run_cli("-d", "-S", "1500", "-o", tmp_path / "part", tmp_path / "a_1.scalcen", "--patterns-bin", PBIN)
    whole = open(tmp_path / "whole_1.fastq", "rb").read()
    parts = [open(tmp_path / f"part.{k}_1.fastq", "rb").read() for k in (1, 2, 3, 4)]
    assert [p.count(b"\n") // 4 for p in parts] == [1500, 1500, 1500, 500]
    assert b"".join(parts) == whole
    O.orc_cli("decompress", PBIN, tmp_path / "a_1.scalcen", tmp_path / "oback")
    assert whole == open(tmp_path / "oback_1.fastq", "rb").read()
    # the C entry directly: streams of the archive in, text out
    ctx = host.Context(0, patterns_bin=open(PBIN, "rb").read())
    r = open(tmp_path / "a_1.scalcer", "rb").read()
    q = open(tmp_path / "a_1.scalceq", "rb").read()
    nm = open(tmp_path / "a_1.scalcen", "rb").read()
    assert r[:8] == b"scalce22" and struct.unpack("<ii", r[8:16]) == (0, L)
    phred = struct.unpack("<q", q[8:16])[0]
    table = np.frombuffer(q[16:16 + 2048000], dtype=np.uint32)
    total = struct.unpack("<Q", q[16 + 2048000:24 + 2048000])[0]
    blocks = torch.frombuffer(bytearray(q[24 + 2048000:]), dtype=torch.uint8).to("cuda:0")
    sym = torch.zeros(total, dtype=torch.uint8, device="cuda:0")
    ctx.ac_decode(table, blocks.data_ptr(), blocks.numel(), total, sym.data_ptr())
    text = ctx.fastq_records(L, r[16:], total // L, sym.data_ptr(), phred, names_payload=nm[9:])
    assert text == whole
    lib = ctx.fastq_records(L, r[16:], total // L, sym.data_ptr(), phred, library="run7")
    recs = lib.split(b"\n")
    assert recs[0] == b"@run7.0" and recs[4 * 4999] == b"@run7.4999" and recs[1] == whole.split(b"\n")[1]


def test_cli_decompress_rejects_truncated_streams(tmp_path):
    """A cut read or name stream ends the reference with '(ERROR) ...' and exit(1) (const.h:77-81); the device path keeps
    that contract -- the directory and name walks on the host see the cut before any kernel runs."""
    n, L = 2000, 100
    synth.write_fastq(str(tmp_path / "in_1.fq"), n, L, seed=43)
    run_cli("-c", "no", "-o", tmp_path / "a", tmp_path / "in_1.fq", "--patterns-bin", PBIN)
    for ext, cut in (("r", 11), ("n", 7)):
        for e in "nrq":
            data = open(tmp_path / f"a_1.scalce{e}", "rb").read()
            open(tmp_path / f"bad_1.scalce{e}", "wb").write(data[:-cut] if e == ext else data)
        r = subprocess.run([CLI, "-d", "-o", str(tmp_path / "x"), str(tmp_path / "bad_1.scalcen"), "--patterns-bin", PBIN],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "(ERROR)" in r.stderr and "truncated" in r.stderr, (ext, r.stderr[-300:])


def test_cli_names_of_every_length(tmp_path):
    """Names from 1 to 60 characters, with and without a comment behind a space (output_name, names.cpp:48-62, stops at
    the first space): the ingest stage keeps names of up to 15 characters in 16-byte cells for the emit stage and goes
    back to the text for longer ones -- both sides of that boundary against the oracle's .scalcen, and back."""
    import numpy as np
    rng = np.random.default_rng(3)
    n, L = 6000, 100
    bases, quals = synth.reads_and_quals(n, L, seed=45)
    alphabet = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789_.:/-", dtype=np.uint8)
    with open(tmp_path / "in_1.fq", "wb") as f:
        for i in range(n):
            ln = 1 + (i % 60) if i < 600 else int(rng.integers(1, 40))
            name = alphabet[rng.integers(0, len(alphabet), size=ln)].tobytes()
            comment = b" length=%d extra" % L if i % 3 == 0 else b""
            f.write(b"@" + name + comment + b"\n" + bases[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")
    run_cli("-c", "no", "-o", tmp_path / "hip", tmp_path / "in_1.fq", "--patterns-bin", PBIN)
    O.orc_cli("compress", PBIN, tmp_path / "in_1.fq", tmp_path / "orc")
    for ext in "nrq":
        assert open(tmp_path / f"hip_1.scalce{ext}", "rb").read() == open(tmp_path / f"orc_1.scalce{ext}", "rb").read(), ext
    run_cli("-d", "-o", tmp_path / "back", tmp_path / "hip_1.scalcen", "--patterns-bin", PBIN)
    O.orc_cli("decompress", PBIN, tmp_path / "orc_1.scalcen", tmp_path / "oback")
    assert open(tmp_path / "back_1.fastq", "rb").read() == open(tmp_path / "oback_1.fastq", "rb").read()


@pytest.mark.parametrize("L", [36, 250, 300])
def test_cli_short_and_long_reads(L, tmp_path):
    """Reads of 250 / 300 bases leave the tiled ingest kernel and the LDS-staged record writer and, from 256 on, carry a
    two-byte `end` (reads.cpp:106-108,130); 36-base reads have barely more key digits than the sort prefix.  Files
    against the oracle's, both ways."""
    n = 3000
    synth.write_fastq(str(tmp_path / "in_1.fq"), n, L, seed=60 + L, n_frac=0.002, dup_frac=0.2)
    run_cli("-c", "no", "-o", tmp_path / "hip", tmp_path / "in_1.fq", "--patterns-bin", PBIN)
    O.orc_cli("compress", PBIN, tmp_path / "in_1.fq", tmp_path / "orc")
    for ext in "nrq":
        assert open(tmp_path / f"hip_1.scalce{ext}", "rb").read() == open(tmp_path / f"orc_1.scalce{ext}", "rb").read(), ext
    run_cli("-d", "-o", tmp_path / "back", tmp_path / "hip_1.scalcen", "--patterns-bin", PBIN)
    O.orc_cli("decompress", PBIN, tmp_path / "orc_1.scalcen", tmp_path / "oback")
    assert open(tmp_path / "back_1.fastq", "rb").read() == open(tmp_path / "oback_1.fastq", "rb").read()


@pytest.mark.parametrize("case", ["se_tiny_pieces", "pe_uneven_mates", "gz_input_two_files", "chunks_across_pieces"])
def test_cli_streams_any_input_in_pieces(case, tmp_path, monkeypatch):
    """The CLI streams its input through pinned chunks (scalce_stream_compress); however the pieces fall -- cut
    mid-record, mates with different bytes per record so that one mate's tail grows, gzip input, several input files,
    -B chunks cut on run-wide sizes -- the archive is the oracle's."""
    n, L = 30000, 100
    paired = case == "pe_uneven_mates"
    files = [tmp_path / "in_1.fq"]
    b1, q1 = synth.reads_and_quals(n, L, seed=51, n_frac=0.003, dup_frac=0.1)
    if paired:
        b2, q2 = synth.reads_and_quals(n, L, seed=52)
        # mate 2's names carry a long comment: its text is a third longer than mate 1's
        open(files[0], "wb").write(synth.fastq_bytes_fast(b1, q1, prefix="p.", suffix="/1"))
        recs = [b"@p.%d/2 " % i + b"c" * 60 + b"\n" + b2[i].tobytes() + b"\n+\n" + q2[i].tobytes() + b"\n" for i in range(n)]
        open(tmp_path / "in_2.fq", "wb").write(b"".join(recs))
    elif case == "gz_input_two_files":
        fq = synth.fastq_bytes_fast(b1, q1)
        cut = fq.index(b"\n@s.17001\n") + 1
        open(files[0], "wb").write(gzip.compress(fq[:cut], 1))
        files.append(tmp_path / "more_1.fq")
        open(files[1], "wb").write(fq[cut:])
        open(tmp_path / "whole_1.fq", "wb").write(fq)
    else:
        open(files[0], "wb").write(synth.fastq_bytes_fast(b1, q1))
    monkeypatch.setenv("SCALCE_PIECE_BYTES", "700000" if case != "se_tiny_pieces" else "40000")
    flags = (["-r"] if paired else []) + (["-B", "1M"] if case == "chunks_across_pieces" else [])
    r = run_cli(*flags, "-c", "no", "-o", tmp_path / "hip", *files, "--patterns-bin", PBIN)
    assert "pieces streamed" in r.stderr
    oin = tmp_path / ("whole_1.fq" if case == "gz_input_two_files" else "in_1.fq")
    O.orc_cli("compress", PBIN, oin, tmp_path / "orc", *[("1048576" if f == "1M" else f) for f in flags])
    for m in ((1, 2) if paired else (1,)):
        for ext in "nrq":
            a = open(tmp_path / f"orc_{m}.scalce{ext}", "rb").read()
            h = open(tmp_path / f"hip_{m}.scalce{ext}", "rb").read()
            assert a == h, f"{case} .scalce{ext} mate {m}: {len(h)} vs {len(a)} bytes"


def test_cli_stream_errors(tmp_path, monkeypatch):
    """A mate that ends early and a text cut inside a record are errors however the pieces fall."""
    n, L = 3000, 60
    b1, q1 = synth.reads_and_quals(n, L, seed=61)
    b2, q2 = synth.reads_and_quals(n, L, seed=62)
    open(tmp_path / "in_1.fq", "wb").write(synth.fastq_bytes_fast(b1, q1, prefix="p.", suffix="/1"))
    fq2 = synth.fastq_bytes_fast(b2[:-5], q2[:-5], prefix="p.", suffix="/2")
    open(tmp_path / "in_2.fq", "wb").write(fq2)
    monkeypatch.setenv("SCALCE_PIECE_BYTES", "50000")
    r = subprocess.run([CLI, "-r", "-c", "no", "-o", str(tmp_path / "o"), str(tmp_path / "in_1.fq"), "--patterns-bin", PBIN],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "(ERROR)" in r.stderr, r.stderr[-300:]
    open(tmp_path / "cut_1.fq", "wb").write(synth.fastq_bytes_fast(b1, q1)[:-7])
    r = subprocess.run([CLI, "-c", "no", "-o", str(tmp_path / "o"), str(tmp_path / "cut_1.fq"), "--patterns-bin", PBIN],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "(ERROR)" in r.stderr, r.stderr[-300:]


@pytest.mark.parametrize("case", ["se_3", "pe_2_uneven_names", "noac_nonames_4"])
def test_cli_gpus_writes_the_archive_of_one_gpu(case, tmp_path, monkeypatch):
    """scalce --gpus N: N processes (here sharing the one GPU over the shared-memory transport), each takes a byte range of
    the input, the ranks write ONE archive with pwrite at computed offsets -- the oracle's archive for the same -B."""
    monkeypatch.setenv("SCALCE_COMM", "shm")
    paired = case.startswith("pe")
    world = int(case.split("_")[1]) if not case.startswith("noac") else 4
    n, L = 60000, 100
    b1, q1 = synth.reads_and_quals(n, L, seed=71, n_frac=0.003, dup_frac=0.1)
    flags = ["-B", "1M"]
    if paired:
        b2, q2 = synth.reads_and_quals(n, L, seed=72)
        open(tmp_path / "in_1.fq", "wb").write(synth.fastq_bytes_fast(b1, q1, prefix="p.", suffix="/1"))
        recs = [b"@p.%d/2 " % i + b"c" * (i % 37) + b"\n" + b2[i].tobytes() + b"\n+\n" + q2[i].tobytes() + b"\n" for i in range(n)]
        open(tmp_path / "in_2.fq", "wb").write(b"".join(recs))
        flags.append("-r")
    else:
        open(tmp_path / "in_1.fq", "wb").write(synth.fastq_bytes_fast(b1, q1))
    if case.startswith("noac"):
        flags += ["-A", "-n", "lib"]
    r = run_cli(*flags, "-c", "no", "--gpus", world, "-o", tmp_path / "hip", tmp_path / "in_1.fq", "--patterns-bin", PBIN)
    assert f"GPUs: {world}" in r.stderr
    O.orc_cli("compress", PBIN, tmp_path / "in_1.fq", tmp_path / "orc", *[("1048576" if f == "1M" else f) for f in flags])
    for m in ((1, 2) if paired else (1,)):
        for ext in "nrq":
            a = open(tmp_path / f"orc_{m}.scalce{ext}", "rb").read()
            h = open(tmp_path / f"hip_{m}.scalce{ext}", "rb").read()
            assert a == h, f"{case} .scalce{ext} mate {m}: {len(h)} vs {len(a)} bytes"


def test_cli_long_reads_two_byte_end_marker(tmp_path):
    """Reads longer than 255 bases store `end` in two bytes (reads.cpp:106-108) and go through the indexed ingest kernels
    (the fused one takes 16..160 bases)."""
    n, L = 3000, 300
    synth.write_fastq(str(tmp_path / "in_1.fq"), n, L, seed=81, n_frac=0.002, dup_frac=0.1)
    run_cli("-c", "no", "-o", tmp_path / "hip", tmp_path / "in_1.fq", "--patterns-bin", PBIN)
    O.orc_cli("compress", PBIN, tmp_path / "in_1.fq", tmp_path / "orc")
    for ext in "nrq":
        assert open(tmp_path / f"orc_1.scalce{ext}", "rb").read() == open(tmp_path / f"hip_1.scalce{ext}", "rb").read(), ext
    run_cli("-d", "-o", tmp_path / "back", tmp_path / "hip_1.scalcen", "--patterns-bin", PBIN)
    O.orc_cli("decompress", PBIN, tmp_path / "orc_1.scalcen", tmp_path / "oback")
    assert open(tmp_path / "back_1.fastq", "rb").read() == open(tmp_path / "oback_1.fastq", "rb").read()


def test_cli_gpus_run_that_B_does_not_cut_is_one_chunk_on_rank_0(tmp_path, monkeypatch):
    """A run shorter than one spill chunk has no chunk boundary for a rank boundary to sit on (compress.cpp:708-715): it is
    ONE chunk, all its rows go to rank 0 (whose order stage then is the merge of every bucket across the ranks), the other
    ranks keep their share of the statistics and of the coder's blocks -- no fallback, no (ERROR), the oracle's archive."""
    monkeypatch.setenv("SCALCE_COMM", "shm")
    n, L = 20000, 100
    b1, q1 = synth.reads_and_quals(n, L, seed=91, n_frac=0.003, dup_frac=0.1)
    open(tmp_path / "in_1.fq", "wb").write(synth.fastq_bytes_fast(b1, q1))
    r = run_cli("-c", "no", "--gpus", 2, "-o", tmp_path / "hip", tmp_path / "in_1.fq", "--patterns-bin", PBIN)   # default -B 4G
    assert "compressing on one GPU" not in r.stderr and "(ERROR)" not in r.stderr, r.stderr[-600:]
    O.orc_cli("compress", PBIN, tmp_path / "in_1.fq", tmp_path / "orc")
    for ext in "nrq":
        assert open(tmp_path / f"orc_1.scalce{ext}", "rb").read() == open(tmp_path / f"hip_1.scalce{ext}", "rb").read(), ext


def test_cli_gpus_failure_of_one_rank_ends_the_run(tmp_path, monkeypatch):
    """A malformed record in ONE rank's byte range: that rank's status travels with the next exchange, every rank stops
    with it, and the command returns 1 -- it once hung, the other ranks waiting in a collective for ever."""
    monkeypatch.setenv("SCALCE_COMM", "shm")
    n, L = 60000, 100
    b1, q1 = synth.reads_and_quals(n, L, seed=92)
    lines = synth.fastq_bytes_fast(b1, q1).split(b"\n")
    lines[4 * 50000 + 1] = lines[4 * 50000 + 1][:-5]          # a short read, in the last rank's part of the file
    open(tmp_path / "bad_1.fq", "wb").write(b"\n".join(lines))
    r = subprocess.run([CLI, "-c", "no", "-B", "1M", "--gpus", "3", "-o", str(tmp_path / "o"), str(tmp_path / "bad_1.fq"),
                        "--patterns-bin", PBIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "(ERROR)" in r.stderr, r.stderr[-600:]


def test_cli_gpus_several_files_gzip_in_and_out(tmp_path, monkeypatch):
    """--gpus with what the reference's one mode takes (compress.cpp:756-811): several input files, one of them gzipped,
    a first file shorter than the quality sample, -c gz.  Against the same command on one GPU (pinned to the reference's
    files by tests/test_ref_files.py::multi) and, where the reference binary is present, against the reference itself."""
    import filecases as F
    monkeypatch.setenv("SCALCE_COMM", "shm")
    n, L = 45000, 100
    b1, q1 = synth.reads_and_quals(n, L, seed=93, n_frac=0.004, dup_frac=0.1)
    b2, q2 = synth.reads_and_quals(n, L, seed=94)
    cuts = [0, 900, 30000, n]
    names = []
    for k in range(3):
        for m, (bb, qq) in enumerate(((b1, q1), (b2, q2))):
            text = b"".join(b"@p.%d/%d\n" % (i, m + 1) + bb[i].tobytes() + b"\n+\n" + qq[i].tobytes() + b"\n" for i in range(cuts[k], cuts[k + 1]))
            fn = tmp_path / ("part%c_%d.fq" % (97 + k, m + 1))
            if k == 1:
                fn = tmp_path / ("part%c_%d.fq.gz" % (97 + k, m + 1))
                text = gzip.compress(text, 1)
            open(fn, "wb").write(text)
        names.append(tmp_path / ("part%c_1.fq%s" % (97 + k, ".gz" if k == 1 else "")))
    flags = ["-r", "-B", "1M", "-s", "5000", "-p", "30", "-c", "gz"]
    run_cli(*flags, "--gpus", 3, "-t", tmp_path / "tmpd", "-o", tmp_path / "multi", *names, "--patterns-bin", PBIN)
    run_cli(*flags, "-o", tmp_path / "one", *names, "--patterns-bin", PBIN)
    have_ref = os.path.exists(F.REF_FULL)
    if have_ref:
        subprocess.run([F.REF_FULL, "compress", PBIN, ",".join(map(str, names)), str(tmp_path / "ref"), "-r", "-B", "1048576", "-s", "5000",
                        "-p", "30", "-c", "gz", "-T", "1", "-t", str(tmp_path / "tmpr")], check=True, capture_output=True)
    for m in (1, 2):
        for ext in "nrq":
            a = maybe_gunzip(tmp_path / f"multi_{m}.scalce{ext}")
            assert a == maybe_gunzip(tmp_path / f"one_{m}.scalce{ext}"), f".scalce{ext} mate {m}: three ranks vs one GPU"
            if have_ref:
                assert a == maybe_gunzip(tmp_path / f"ref_{m}.scalce{ext}"), f".scalce{ext} mate {m}: three ranks vs the reference"
    assert open(tmp_path / "multi_1.scalcer", "rb").read(2) == b"\x1f\x8b"
    assert not list((tmp_path / "tmpd").glob("scalce_gpus_*")), "the plain copy of the input was left behind"


def test_cli_input_files_without_a_trailing_newline(tmp_path, monkeypatch):
    """Several input files none of which ends in a newline: every file's last quality line still ends with the file
    (concatenated byte-wise it ran into the next file's '@name').  The reference itself cannot read such a file
    (it takes the read length from a line it assumes ends in a newline: 'read length: 99', then fails), so the pin is the
    archive of the same records WITH the newlines -- on one GPU and through --gpus (the plain copy under -t)."""
    monkeypatch.setenv("SCALCE_COMM", "shm")
    n, L = 30000, 100
    b1, q1 = synth.reads_and_quals(n, L, seed=96, n_frac=0.004, dup_frac=0.1)
    cuts = [0, 700, 20000, n]
    with_nl, without = [], []
    for k in range(3):
        text = b"".join(b"@s.%d\n" % i + b1[i].tobytes() + b"\n+\n" + q1[i].tobytes() + b"\n" for i in range(cuts[k], cuts[k + 1]))
        a, b = tmp_path / ("nl%c_1.fq" % (97 + k)), tmp_path / ("no%c_1.fq" % (97 + k))
        open(a, "wb").write(text)
        open(b, "wb").write(gzip.compress(text[:-1], 1) if k == 1 else text[:-1])
        with_nl.append(a)
        without.append(b)
    flags = ["-c", "no", "-B", "1M", "-s", "5000"]
    run_cli(*flags, "-o", tmp_path / "want", *with_nl, "--patterns-bin", PBIN)
    run_cli(*flags, "-o", tmp_path / "got", *without, "--patterns-bin", PBIN)
    run_cli(*flags, "--gpus", 2, "-t", tmp_path / "tmpd", "-o", tmp_path / "got2", *without, "--patterns-bin", PBIN)
    for ext in "nrq":
        want = open(tmp_path / f"want_1.scalce{ext}", "rb").read()
        assert open(tmp_path / f"got_1.scalce{ext}", "rb").read() == want, ext
        assert open(tmp_path / f"got2_1.scalce{ext}", "rb").read() == want, ext + " (--gpus 2)"
    # --gpus: a temporary directory that cannot be made is an error, not a silent /tmp
    r = subprocess.run([CLI, *flags, "--gpus", "2", "-t", str(tmp_path / "no" / "such" / "dir"), "-o", str(tmp_path / "x"), *map(str, without),
                        "--patterns-bin", PBIN], capture_output=True, text=True)
    assert r.returncode != 0 and "temporary directory" in r.stderr, r.stderr[-400:]


def test_cli_gpus_over_rccl_when_there_are_two_gpus(tmp_path):
    """The N > 1 data path over RCCL (ncclAllGather / AllReduce / Send / Recv between processes that own a GPU each): runs
    wherever two GPUs are visible -- the driver's 8-GPU box -- and is skipped on a one-GPU box, where the same code runs
    over the shared-memory transport in the tests above."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    n, L = 200000, 100
    b1, q1 = synth.reads_and_quals(n, L, seed=95, n_frac=0.003, dup_frac=0.1)
    open(tmp_path / "in_1.fq", "wb").write(synth.fastq_bytes_fast(b1, q1))
    env = dict(os.environ)
    env.pop("SCALCE_COMM", None)
    r = subprocess.run([CLI, "-c", "no", "-B", "4M", "--gpus", "2", "-o", str(tmp_path / "hip"), str(tmp_path / "in_1.fq"),
                        "--patterns-bin", PBIN], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "GPUs: 2" in r.stderr, r.stderr[-800:]
    O.orc_cli("compress", PBIN, tmp_path / "in_1.fq", tmp_path / "orc", "-B", str(4 << 20))
    for ext in "nrq":
        assert open(tmp_path / f"orc_1.scalce{ext}", "rb").read() == open(tmp_path / f"hip_1.scalce{ext}", "rb").read(), ext
    # and the bench's sharded path at two ranks (torchrun, RCCL): one JSON line, positive throughput
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29741", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--reads", "2000000", "--cpu-sample", "0", "--no-e2e"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    import json
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
