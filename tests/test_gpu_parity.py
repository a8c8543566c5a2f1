"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Bit-exact everywhere: tokens (core, end) incl. the cumulative tie-break, output permutation, 2-bit
records, names, q' symbols, trigram table, arithmetic-coder bytes, and finally the decompressed FASTQ.
"""
import hashlib
import os
import struct

import numpy as np
import pytest

import oraclelib as O
from scalce_amd import format as fmt
from scalce_amd import host, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx(patterns_blob):
    return host.Context(0, patterns_bin=patterns_blob)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_against_oracle(ctx, trie, bases, quals, qoff=33, qvals=None, label=""):
    from gpu_util import hip_compress, oracle_streams
    n, L = bases.shape
    fq = synth.fastq_bytes_fast(bases, quals)
    qm = None if qvals is None else [(qoff, qvals), (qoff, qvals)]
    b = hip_compress(ctx, fq, L, qmap=qm)
    ref = oracle_streams(trie, bases, quals, qoff, qvals)
    st = b.stats()
    tok = b.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)
    bad = np.flatnonzero((tok[:, 0] != ref["pat"]) | (tok[:, 1] != ref["end"]))
    assert len(bad) == 0, f"{label}: {len(bad)} token mismatches, first at read {bad[:5]}: hip {tok[bad[:5]]} " \
                          f"oracle {ref['pat'][bad[:5]]},{ref['end'][bad[:5]]} stats {st}"
    perm = b.output(host.OUT_PERM, 0, np.uint32)
    badp = np.flatnonzero(perm != ref["perm"])
    assert len(badp) == 0, f"{label}: permutation differs at {len(badp)} of {n} positions, first {badp[:5]}"
    qin = b.output(host.OUT_QINPUT, 0).reshape(n, L)
    assert (qin == ref["qp"]).all(), f"{label}: q' mismatch"
    f4 = b.output(host.OUT_FREQ4, 0, np.uint64)
    assert (f4 + 1 == ref["f4"]).all(), f"{label}: trigram table mismatch at {np.flatnonzero(f4 + 1 != ref['f4'])[:5]}"
    table = b.output(host.OUT_TABLE, 0, np.uint32)
    assert (table == O.ac_scale(ref["f4"], 1)).all(), f"{label}: scaled table mismatch"
    qs = b.output(host.OUT_QSTREAM, 0)
    want_qs = ref["qp"][ref["perm"]].reshape(-1)
    assert (qs == want_qs).all(), f"{label}: reordered quality stream mismatch"
    enc = b.output(host.OUT_QUAL, 0)
    want = O.AcStat(table).encode_stream(want_qs)
    assert len(enc) == len(want), f"{label}: AC length {len(enc)} vs {len(want)}"
    neq = np.flatnonzero(enc != want)
    assert len(neq) == 0, f"{label}: AC bytes differ first at {neq[:5]} of {len(want)}"
    return b, ref, st


@pytest.mark.parametrize("name", ["se100", "se100_lossy", "se150_text", "se36_ties"])
def test_golden_cases(name, patterns_blob):
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    kw = eval(str(g["kw"]))  # noqa: S307
    bases, quals = synth.reads_and_quals(int(g["n"]), int(g["L"]), seed=int(g["seed"]), **kw)
    if "ptxt" in g:
        txt = str(g["ptxt"]).encode()
        c, trie = host.Context(0, patterns_text=txt), O.Trie(text=txt)
    else:
        c, trie = host.Context(0, patterns_bin=patterns_blob), O.Trie(blob=patterns_blob)
    qoff, qvals = (int(g["lut"][0]), g["lut"][1:]) if "lut" in g else (33, None)
    b, ref, st = check_against_oracle(c, trie, bases, quals, qoff, qvals, label=name)
    print(name, st)
    # and directly against the vectors produced by the reference objects
    tok = b.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)
    assert sha(tok) == str(g["sha_tok"])
    assert sha(b.output(host.OUT_PERM, 0, np.uint32).astype(np.int64)) == str(g["sha_order"])
    assert sha(b.output(host.OUT_QINPUT, 0)) == str(g["sha_qual"])
    assert sha(b.output(host.OUT_FREQ4, 0, np.uint64) + 1) == str(g["sha_freq4"])
    assert sha(b.output(host.OUT_QUAL, 0)) == str(g["sha_ac"])
    assert sha(names_in_input_order(b)) == str(g["sha_names"])   # output_name (names.cpp:48-62) of the reference itself
    # .scalcer records (headers stripped) == concatenated output_read bytes in emission order
    lens = trie.pattern_lens()
    reads = b.output(host.OUT_READS, 0)
    perm, pos, packed_in_order = ref["perm"], 0, []
    L = int(g["L"])
    goldp = g["packed"]
    offs = np.zeros(len(bases) + 1, dtype=np.int64)
    for r in range(len(bases)):
        lvl = int(lens[ref["pat"][r]]) if ref["pat"][r] >= 0 else 0
        offs[r + 1] = offs[r] + (L - lvl + 3) // 4
    k = 0
    while k < len(perm):
        core, cnt = struct.unpack_from("<iq", reads, pos)
        pos += 12
        p = ref["pat"][perm[k]]
        assert core == (p if p >= 0 else host.ROOT_CORE)
        for i in range(cnt):
            r = perm[k + i]
            nb = offs[r + 1] - offs[r]
            assert (reads[pos:pos + nb] == goldp[offs[r]:offs[r + 1]]).all(), f"record {k + i}"
            assert reads[pos + nb] == ref["end"][r]
            pos += nb + 1
        k += cnt
    assert pos == len(reads)


def names_in_input_order(b):
    """SCALCE_OUT_NAMES holds [u8 n][bytes] per record in emission order; put the records back into input order."""
    names = b.output(host.OUT_NAMES, 0)
    perm = b.output(host.OUT_PERM, 0, np.uint32)
    n = len(perm)
    start = np.zeros(n + 1, dtype=np.int64)
    pos = 0
    for k in range(n):
        start[k] = pos
        pos += 1 + int(names[pos])
    start[n] = pos
    assert pos == len(names)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    return np.concatenate([names[start[inv[r]]:start[inv[r] + 1]] for r in range(n)])


@pytest.mark.parametrize("name", ["se100_f7", "pe150", "pe150_lossy_f2"])
def test_golden_paired_and_shrink_factor(name, ctx):
    """Vectors from the reference's own objects for what every GPU config of BASELINE.json runs into and the round-1
    tests never did: the shrink factor of compress.cpp:297-313 (> 1 from 43 M x 100 symbols on: C2 2, C3 7, C4 24) and
    mate 2 of a paired run (output_read(.., 0, 0), output_quality(.., ZZ = 1) with its own prev[] and counters)."""
    import torch
    from gpu_util import device_bytes
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    kw = eval(str(g["kw"]))  # noqa: S307
    n, L, factor = int(g["n"]), int(g["L"]), int(g["factor"])
    paired = "seed2" in g
    bases, quals = synth.reads_and_quals(n, L, seed=int(g["seed"]), **kw)
    lut = lambda key: (int(g[key][0]), g[key][1:]) if key in g else (33, np.arange(128))  # noqa: E731
    qm = [lut("lut"), lut("lut2") if paired else lut("lut")]
    if paired:
        bases2, quals2 = synth.reads_and_quals(n, L, seed=int(g["seed2"]), **kw)
        fq1 = synth.fastq_bytes_fast(bases, quals, prefix="p.", suffix="/1")
        fq2 = synth.fastq_bytes_fast(bases2, quals2, prefix="p.", suffix="/2")
    else:
        fq1, fq2 = synth.fastq_bytes_fast(bases, quals), None
    t1 = device_bytes(fq1)
    t2 = device_bytes(fq2) if paired else None
    b = host.Batch(ctx, L, n + 8, max(len(fq1), len(fq2 or b"")) + 64, paired=paired, read_len2=L, qmap=qm)
    b.front(t1.data_ptr(), len(fq1), t2.data_ptr() if paired else None, len(fq2 or b""))
    b.finish()
    assert sha(b.output(host.OUT_TOKENS, 0, np.int32)) == str(g["sha_tok"])
    perm = b.output(host.OUT_PERM, 0, np.uint32)
    assert sha(perm.astype(np.int64)) == str(g["sha_order"])
    assert sha(names_in_input_order(b)) == str(g["sha_names"])
    nm = 2 if paired else 1
    tables = torch.zeros(2 * 512000, dtype=torch.int32, device="cuda:0")
    for m in range(nm):
        key = (lambda k: k if m == 0 else ("sha_freq4_2" if k == "sha_freq4" else k + "2"))  # golden keys of mate 2
        assert sha(b.output(host.OUT_QINPUT, m)) == str(g[key("sha_qual")]), m
        assert sha(b.output(host.OUT_FREQ4, m, np.uint64) + 1) == str(g[key("sha_freq4")]), m
        f4p, _ = b.output_ptr(host.OUT_FREQ4, m)
        ctx.ac_scale(f4p, factor, tables.data_ptr() + 4 * 512000 * m)
        torch.cuda.synchronize()
        tab = tables[512000 * m:512000 * (m + 1)].cpu().numpy().view(np.uint32)
        assert (tab == O.ac_scale(b.output(host.OUT_FREQ4, m, np.uint64) + 1, factor)).all(), f"mate {m + 1}: scaled table vs oracle"
        assert sha(tab) == str(g[key("sha_table")]), f"mate {m + 1}: scaled table (factor {factor}) vs the reference's"
    b.entropy(tables.data_ptr())
    b.finish()
    for m in range(nm):
        key = (lambda k: k if m == 0 else k + "2")
        enc = b.output(host.OUT_QUAL, m)
        assert len(enc) == int(g["ac_len" if m == 0 else "ac2_len"])
        assert sha(enc) == str(g[key("sha_ac")]), f"mate {m + 1}: coder bytes against the scaled table"
    if paired:  # mate 2's .scalcer payload: bare whole-read records in mate 1's order
        r2 = b.output(host.OUT_READS, 1).reshape(n, (L + 3) // 4)
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        assert sha(r2[inv]) == str(g["sha_packed2"])


def test_two_ac_blocks_and_decode(ctx, oracle_trie):
    g = np.load(os.path.join(GOLD, "se100_110k.npz"), allow_pickle=False)
    bases, quals = synth.reads_and_quals(int(g["n"]), int(g["L"]), seed=int(g["seed"]))
    b, ref, st = check_against_oracle(ctx, oracle_trie, bases, quals, label="110k")
    print("110k", st)
    assert sha(b.output(host.OUT_QUAL, 0)) == str(g["sha_ac"])
    assert sha(b.output(host.OUT_PERM, 0, np.uint32).astype(np.int64)) == str(g["sha_order"])
    # GPU decoder inverts the GPU encoder
    import torch
    nsym = bases.size
    out = torch.zeros(nsym, dtype=torch.uint8, device="cuda:0")
    p, nbytes = b.output_ptr(host.OUT_QUAL, 0)
    ctx.ac_decode(b.output(host.OUT_TABLE, 0, np.uint32), p, nbytes, nsym, out.data_ptr())
    assert (out.cpu().numpy() == b.output(host.OUT_QSTREAM, 0)).all()


@pytest.mark.parametrize("case", ["plain", "lossy", "noac", "nonames", "paired", "chunks", "odd_len"])
def test_files_match_oracle_and_roundtrip(case, tmp_path, ctx, patterns_blob):
    """Byte-identical .scalce{n,r,q} versus the oracle's writer, and the oracle's decompressor
    restores the FASTQ from the files written from the GPU streams."""
    from gpu_util import hip_compress
    pbin = os.path.join(GOLD, "patterns.bin")
    L = 75 if case == "odd_len" else 100
    n = 6000
    paired = case == "paired"
    b1, q1 = synth.write_fastq(str(tmp_path / "in_1.fq"), n, L, seed=21, n_frac=0.004, dup_frac=0.15,
                               paired_suffix="/1" if paired else None)
    fq1 = open(tmp_path / "in_1.fq", "rb").read()
    fq2 = None
    if paired:
        synth.write_fastq(str(tmp_path / "in_2.fq"), n, L, seed=22, paired_suffix="/2")
        fq2 = open(tmp_path / "in_2.fq", "rb").read()
    extra, kw = [], {}
    lossy = 30 if case == "lossy" else 0
    off, vals, Ls = fmt.sample_qmap(fq1, lossy=lossy)
    assert Ls == L
    qm = [(off, vals)]
    if paired:
        off2, vals2, _ = fmt.sample_qmap(fq2, lossy=lossy)
        qm.append((off2, vals2))
        extra.append("-r")
    else:
        qm.append((off, vals))
    if lossy:
        extra += ["-p", "30"]
    if case == "noac":
        extra.append("-A"); kw["no_ac"] = True
    if case == "nonames":
        extra += ["-n", "lib"]; kw["use_names"] = False
    if case == "chunks":
        extra += ["-B", "200000"]; kw["bucket_set_size"] = 200000
    O.orc_cli("compress", pbin, tmp_path / "in_1.fq", tmp_path / "orc", *extra)
    b = hip_compress(ctx, fq1, L, fastq2=fq2, L2=L, qmap=qm, **kw)
    if case == "chunks":
        assert b.stats()["chunks"] > 1
    fmt.write_archive(str(tmp_path / "hip"), b, off, library="lib")
    for m in ((1, 2) if paired else (1,)):
        for ext in "nrq":
            a = open(tmp_path / f"orc_{m}.scalce{ext}", "rb").read()
            h = open(tmp_path / f"hip_{m}.scalce{ext}", "rb").read()
            assert len(a) == len(h), f"{case} .scalce{ext} mate {m}: {len(h)} vs oracle {len(a)} bytes"
            assert a == h, f"{case} .scalce{ext} mate {m} differs at byte {next(i for i in range(len(a)) if a[i] != h[i])}"
    dextra = (["-r"] if paired else []) + (["-n", "lib"] if case == "nonames" else [])
    O.orc_cli("decompress", pbin, tmp_path / "hip_1.scalcen", tmp_path / "back", *dextra)
    if not lossy and case != "nonames":
        for m in ((1, 2) if paired else (1,)):
            src = open(tmp_path / f"in_{m}.fq", "rb").read().split(b"\n")
            got = open(tmp_path / f"back_{m}.fastq", "rb").read().split(b"\n")

            def canon(Ls_):
                return sorted((Ls_[i], Ls_[i + 1], Ls_[i + 2], bytes(33 if x == 78 else y for x, y in zip(Ls_[i + 1], Ls_[i + 3])))
                              for i in range(0, len(Ls_) - 1, 4))
            assert canon(src) == canon(got)


def test_ac_step_closed_form_selftest(ctx):
    """64 M random + crafted coder states: reciprocal multiply-high and the merged shift reproduce the
    reference's division and bit-by-bit renormalisation loop exactly (incl. R+1 wrap, last symbol,
    all-32-bits-agree, long underflow runs)."""
    for general in (0, 1, 2):  # 0 production step, 1 all-states step, 2 plain-round step with its fallback
        for seed in (1, 2, 3, 4):
            out = ctx.selftest_ac(1 << 24, seed, general)
            assert out[0] == 0, (f"general={general} seed {seed}: {out[0]} mismatches, first lo={out[1]:#x} hi={out[2]:#x} "
                                 f"c_lo={out[3]} c_hi={out[4]} d={out[5]}")


def test_ac_slow_emit_path(ctx, oracle_trie, monkeypatch):
    """Force every pending-underflow round through the serial emit path (normally taken about once per 2^32
    symbols) and require the same bytes."""
    monkeypatch.setenv("SCALCE_AC_SLOW_THRESHOLD", "0")
    bases, quals = synth.reads_and_quals(30000, 100, seed=77)
    check_against_oracle(ctx, oracle_trie, bases, quals, label="slow-emit")


def test_ac_general_step_same_bytes(ctx, oracle_trie, monkeypatch):
    """The all-states coder step (selected when a context total exceeds 2^30) gives the same stream."""
    monkeypatch.setenv("SCALCE_AC_GENERAL", "1")
    bases, quals = synth.reads_and_quals(30000, 100, seed=78)
    check_against_oracle(ctx, oracle_trie, bases, quals, label="general-step")


def test_heavy_duplicates_two_phase_order(ctx, oracle_trie, monkeypatch):
    """Few distinct reads, many copies, and long shared prefixes: nearly every record goes through the second
    sort phase (runs that tie on bucket + 16-base prefix); same permutation as the all-digits sort."""
    rng = np.random.default_rng(5)
    base_reads, _ = synth.reads_and_quals(60, 100, seed=41)
    idx = rng.integers(0, 60, 20000)
    bases = base_reads[idx].copy()
    mut = rng.random(20000) < 0.5          # half of them differ only near the end of the read
    pos = rng.integers(70, 100, 20000)
    bases[mut, pos[mut]] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, mut.sum())]
    _, quals = synth.reads_and_quals(20000, 100, seed=42)
    b, ref, st = check_against_oracle(ctx, oracle_trie, bases, quals, label="dups")
    assert st["order_run_members"] > 15000
    monkeypatch.setenv("SCALCE_ORDER_SINGLE_PHASE", "1")
    b2, _, st2 = check_against_oracle(ctx, oracle_trie, bases, quals, label="dups-single-phase")
    assert st2["order_run_members"] == 0


def test_paired_150bp_mid_size(ctx, patterns_blob, tmp_path):
    """BASELINE configs[2] shape (150 bp paired-end, -r) at 400 k pairs: files equal the oracle's, both mates."""
    from gpu_util import hip_compress
    n, L = 400000, 150
    b1, q1 = synth.write_fastq(str(tmp_path / "in_1.fq"), n, L, seed=61, n_frac=0.001, paired_suffix="/1")
    synth.write_fastq(str(tmp_path / "in_2.fq"), n, L, seed=62, n_frac=0.001, paired_suffix="/2")
    fq1 = open(tmp_path / "in_1.fq", "rb").read()
    fq2 = open(tmp_path / "in_2.fq", "rb").read()
    off, vals, _ = fmt.sample_qmap(fq1)
    off2, vals2, _ = fmt.sample_qmap(fq2)
    O.orc_cli("compress", os.path.join(GOLD, "patterns.bin"), tmp_path / "in_1.fq", tmp_path / "orc", "-r")
    b = hip_compress(ctx, fq1, L, fastq2=fq2, L2=L, qmap=[(off, vals), (off2, vals2)])
    fmt.write_archive(str(tmp_path / "hip"), b, off)
    for m in (1, 2):
        for ext in "nrq":
            assert open(tmp_path / f"orc_{m}.scalce{ext}", "rb").read() == open(tmp_path / f"hip_{m}.scalce{ext}", "rb").read(), (m, ext)


def test_large_shard_properties(ctx):
    """3 M reads (beyond what the oracle is asked to redo here): size-independent properties -- the order is a
    permutation sorted by (bucket, key), bucket counts add up, the GPU decoder inverts every coded block, and the
    .scalcer payload has exactly the advertised size."""
    import torch
    from scalce_amd import synth_gpu
    n, L = 3_000_000, 100
    text = synth_gpu.fastq_on_device(n, L, torch.device("cuda", 0), seed=123)
    b = host.Batch(ctx, L, n + 8, text.numel() + 64)
    b.compress(text.data_ptr(), text.numel())
    b.finish()
    assert b.n_reads == n
    perm = b.output(host.OUT_PERM, 0, np.uint32)
    assert (np.sort(perm) == np.arange(n, dtype=np.uint32)).all()
    tok = b.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)
    counts = b.output(host.OUT_BUCKET_COUNTS, 0, np.uint64)
    assert counts.sum() == n
    order = ctx.bucket_patterns()
    rank_of_pattern = {int(p): i for i, p in enumerate(order[:-1])}
    bucket = np.array([rank_of_pattern[p] if p >= 0 else len(order) - 1 for p in tok[perm, 0]])
    assert (np.diff(bucket) >= 0).all(), "buckets not in emission order"
    assert (np.bincount(bucket, minlength=len(order)) == counts.astype(np.int64)).all()
    # coreless reads have end 0, others end >= core length
    lens = np.array([len(ctx.pattern(int(p))) for p in order[:-1]] + [0])
    assert ((tok[:, 1] == 0) == (tok[:, 0] < 0)).all() and (tok[perm, 1] >= lens[bucket]).all()
    # decoder round trip on the device
    nsym = n * L
    out = torch.zeros(nsym, dtype=torch.uint8, device="cuda:0")
    p, nbytes = b.output_ptr(host.OUT_QUAL, 0)
    ctx.ac_decode(b.output(host.OUT_TABLE, 0, np.uint32), p, nbytes, nsym, out.data_ptr())
    qs_ptr, qs_n = b.output_ptr(host.OUT_QSTREAM, 0)
    want = torch.empty(nsym, dtype=torch.uint8, device="cuda:0")
    ctx.copy_d2d(want.data_ptr(), qs_ptr, qs_n)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    recsz = (L - lens + 3) // 4 + 1
    assert len(b.output(host.OUT_READS, 0)) == int((counts.astype(np.int64) * recsz).sum() + 12 * (counts > 0).sum())


def test_full_size_shard_properties(ctx):
    """BASELINE.json configs[1] at its full size, 50 M x 100 bp (10.8 GB): size-independent properties checked on the
    device -- the order is a permutation, buckets come in emission order with the advertised counts, cores end inside
    the read, the GPU decoder inverts all 477 coded blocks back to the reordered quality stream, the payload sizes are
    exactly the advertised ones.  (The oracle redoes this path at 1.5 M reads in bench.py and at up to 230 k here.)"""
    import torch
    from scalce_amd import synth_gpu
    dev = torch.device("cuda", 0)
    n, L = 50_000_000, 100
    text = synth_gpu.fastq_on_device(n, L, dev, seed=20261003)
    b = host.Batch(ctx, L, n + 8, text.numel() + 64)
    b.compress(text.data_ptr(), text.numel())
    b.finish()
    assert b.n_reads == n

    def dev_array(which, dtype, count):
        ptr, nb = b.output_ptr(which, 0)
        t = torch.empty(count, dtype=dtype, device=dev)
        assert nb == t.numel() * t.element_size()
        ctx.copy_d2d(t.data_ptr(), ptr, nb)
        torch.cuda.synchronize()
        return t

    perm = dev_array(host.OUT_PERM, torch.int32, n).to(torch.int64)   # values < 2^31
    seen = torch.zeros(n, dtype=torch.bool, device=dev)
    seen[perm] = True
    assert bool(seen.all()), "order is not a permutation"
    del seen
    tok = dev_array(host.OUT_TOKENS, torch.int32, 2 * n).view(n, 2)
    counts = b.output(host.OUT_BUCKET_COUNTS, 0, np.uint64).astype(np.int64)
    assert counts.sum() == n
    order = ctx.bucket_patterns()
    nb1 = len(order)
    rank = torch.full((ctx.n_patterns + 1,), nb1 - 1, dtype=torch.int64, device=dev)  # last slot: no core (-1)
    rank[torch.from_numpy(order[:-1].astype(np.int64)).to(dev)] = torch.arange(nb1 - 1, device=dev)
    bucket = rank[tok[:, 0].to(torch.int64)[perm]]          # index -1 -> last slot
    assert bool((bucket[1:] >= bucket[:-1]).all()), "buckets not in emission order"
    assert (torch.bincount(bucket, minlength=nb1).cpu().numpy() == counts).all()
    lens = torch.from_numpy(np.array([len(ctx.pattern(int(p))) for p in order[:-1]] + [0], dtype=np.int64)).to(dev)
    end = tok[:, 1].to(torch.int64)[perm]
    assert bool(((end == 0) == (bucket == nb1 - 1)).all()) and bool((end >= lens[bucket]).all()) and bool((end <= L).all())
    del bucket, end, perm, tok
    # decoder round trip of the whole coded stream on the device
    nsym = n * L
    out = torch.zeros(nsym, dtype=torch.uint8, device=dev)
    p, nbytes = b.output_ptr(host.OUT_QUAL, 0)
    ctx.ac_decode(b.output(host.OUT_TABLE, 0, np.uint32), p, nbytes, nsym, out.data_ptr())
    want = dev_array(host.OUT_QSTREAM, torch.uint8, nsym)
    assert torch.equal(out, want), "decoded stream differs from the reordered quality stream"
    # the bench workload runs at shrink factor 2 (5e9 symbols): the table the coder used against the oracle's scaling
    f4 = b.output(host.OUT_FREQ4, 0, np.uint64)
    assert int(f4.sum()) == nsym - 2
    factor = 1 + nsym // 0xFFFFFFFF
    assert factor == 2
    assert (b.output(host.OUT_TABLE, 0, np.uint32) == O.ac_scale(f4 + 1, factor)).all(), "scaled table at factor 2"
    recsz = (L - lens.cpu().numpy() + 3) // 4 + 1
    assert b.output_ptr(host.OUT_READS, 0)[1] == int((counts * recsz).sum() + 12 * (counts > 0).sum())
    assert b.output_ptr(host.OUT_NAMES, 0)[1] == int(b.output(host.OUT_NAMELEN, 0).astype(np.int64).sum()) + n


def test_malformed_input_is_an_error(ctx):
    from gpu_util import device_bytes
    b1, q1 = synth.reads_and_quals(50, 40, seed=3)
    fq = bytearray(synth.fastq_bytes_fast(b1, q1))
    bad = bytes(fq[:200]) + b"ACGT\n" + bytes(fq[200:])  # breaks the 4-line structure
    t = device_bytes(bad)
    b = host.Batch(ctx, 40, 100, len(bad) + 64)
    with pytest.raises(host.ScalceError):
        b.compress(t.data_ptr(), len(bad))
        b.finish()
    # one read shorter than the others (the reference exits: compress.cpp:628-634)
    lines = bytes(fq).split(b"\n")
    lines[5] = lines[5][:-1]
    lines[7] = lines[7][:-1]
    bad2 = b"\n".join(lines)
    t2 = device_bytes(bad2)
    b2 = host.Batch(ctx, 40, 100, len(bad2) + 64)
    with pytest.raises(host.ScalceError):
        b2.compress(t2.data_ptr(), len(bad2))
        b2.finish()
    # a quality character 80 or more above the offset: the coder's tables have no row for it (arithmetic.h:47)
    for L in (40, 41):
        b3_, q3 = synth.reads_and_quals(3000, L, seed=4)
        q3[2345, L - 1] = 33 + 85
        fq3 = synth.fastq_bytes_fast(b3_, q3)
        t3 = device_bytes(fq3)
        b3 = host.Batch(ctx, L, 3100, len(fq3) + 64, qmap=[(33, np.arange(128, dtype=np.int32))])
        with pytest.raises(host.ScalceError, match="2345|symbol"):
            b3.compress(t3.data_ptr(), len(fq3))
            b3.finish()


def test_empty_shard(ctx):
    import torch
    t = torch.zeros(16, dtype=torch.uint8, device="cuda:0")
    b = host.Batch(ctx, 100, 16, 1024)
    b.compress(t.data_ptr(), 0)
    b.finish()
    assert b.n_reads == 0
    assert len(b.output(host.OUT_READS, 0)) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("alphabet", ["full_span", "single", "gaps", "top_only"])
def test_quality_statistics_on_odd_alphabets(alphabet, ctx, oracle_trie):
    """The trigram table is built in LDS slices laid out over the span of the symbols that occur
    (trigram_pass_k): 2 passes for a usual alphabet, 20 when q' spans all 80 values.  Whole path against the
    oracle on alphabets the generator never produces, so every pass count and both ends of the table are hit."""
    rng = np.random.default_rng(11)
    n, L = 6000, 100
    bases, _ = synth.reads_and_quals(n, L, seed=12)
    if alphabet == "full_span":      # q' 0..79: 4 leading symbols per pass, 20 passes
        q = rng.integers(0, 80, size=(n, L))
    elif alphabet == "single":       # one symbol: A = 1
        q = np.full((n, L), 37)
    elif alphabet == "gaps":         # lossy-style alphabet with holes, low and high ends far apart
        q = rng.choice(np.array([0, 3, 30, 41, 77]), size=(n, L))
    else:                            # only the last symbols of the table
        q = rng.integers(76, 80, size=(n, L))
    quals = (q + 33).astype(np.uint8)
    check_against_oracle(ctx, oracle_trie, bases, quals, label=alphabet)


@pytest.mark.gpu
def test_grouped_coder_launch_equals_separate_launches(ctx, monkeypatch):
    """scalce_batch_entropy_begin_group: ONE coder launch (four blocks per workgroup, per-block descriptors) over three
    shards of different sizes and alphabets -- so workgroups hold blocks of different shards, tables and lengths --
    gives every shard the bytes its own launch gives it (one-block and four-block kernel)."""
    from gpu_util import device_bytes
    rng = np.random.default_rng(5)
    specs = [(230_000, 100, 31), (120_000, 100, 32), (70_000, 36, 33)]   # 3, 2 and 1 blocks, the last ones short
    texts, alone = [], []
    for n, L, seed in specs:
        bases, quals = synth.reads_and_quals(n, L, seed=seed)
        if seed == 32:
            quals = (rng.integers(0, 80, size=(n, L)) + 33).astype(np.uint8)
        fq = synth.fastq_bytes_fast(bases, quals)
        texts.append((device_bytes(fq), len(fq), n, L))
    for variant in ("1", "4", "64"):
        monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", variant)
        outs = []
        for t, nb, n, L in texts:
            b = host.Batch(ctx, L, n + 8, nb + 64)
            b.compress(t.data_ptr(), nb)
            b.finish()
            outs.append(b.output(host.OUT_QUAL, 0).copy())
        alone.append(outs)
    monkeypatch.delenv("SCALCE_AC_BLOCKS_PER_WG")
    for a1, a4, a64 in zip(*alone):
        assert len(a1) == len(a4) and (a1 == a4).all()
        assert len(a1) == len(a64) and (a1 == a64).all()
    group = []
    for t, nb, n, L in texts:
        b = host.Batch(ctx, L, n + 8, nb + 64)
        b.front(t.data_ptr(), nb)
        group.append(b)
    host.entropy_begin_group(group)
    for b, want in zip(group, alone[0]):
        b.finish()
        got = b.output(host.OUT_QUAL, 0)
        assert len(got) == len(want) and (got == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("blocks_per_wave", ["1", "4", "8", "64"])
def test_rows_coder_redo_path(blocks_per_wave, ctx, oracle_trie, monkeypatch):
    """The coder kernels find a step that needed the general path by the absorbing state it leaves behind (range = 2^32
    -> M = 0 in the last lane at the end of the super-round) and redoes the block's super-round on the general path.
    SCALCE_AC_TEST_POISON makes every third super-round pretend that happened: state restored, outcomes rewritten in
    the raw format, same bytes as the oracle."""
    monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", blocks_per_wave)
    monkeypatch.setenv("SCALCE_AC_TEST_POISON", "3")
    bases, quals = synth.reads_and_quals(150_000, 100, seed=77)   # two blocks, the second one short
    check_against_oracle(ctx, oracle_trie, bases, quals, label="poison" + blocks_per_wave)


@pytest.mark.parametrize("case", ["narrow", "rare_outside", "wide", "top_symbol", "plain_kernel", "narrow_16_chains", "rare_outside_8_chains"])
def test_decoder_compact_rows_and_lds_cache(case, ctx, monkeypatch):
    """scalce_ac_decode keeps compact rows (upper bounds of the symbols that occur) and the hot contexts' rows in LDS.
    Crafted tables drive every path: a narrow alphabet (all contexts cached), symbols whose count is 1 everywhere but
    which still occur (full-row path), an alphabet wider than one pass (the plain kernel), symbol 79 (the context's
    last interval).  Encoder and decoder are both checked against the oracle's coder with the same table."""
    import torch
    if case.endswith("_chains"):   # the workgroup shapes of archives with thousands of blocks, on three
        monkeypatch.setenv("SCALCE_AC_DECODE_WPB", case.split("_")[-2])
        case = case.rsplit("_", 2)[0]
    rng = np.random.default_rng(77 + len(case))
    nsym = 2 * 10 * 1024 * 1024 + 12345  # three blocks, the last one short
    if case in ("narrow", "plain_kernel"):
        alphabet, weights = np.array([2, 11, 25, 37]), np.array([0.1, 0.2, 0.3, 0.4])
    elif case == "rare_outside":
        alphabet, weights = np.arange(20, 42), None
    elif case == "wide":
        alphabet, weights = np.arange(0, 80), None
    else:
        alphabet, weights = np.array([30, 31, 40, 78, 79]), None
    sym = rng.choice(alphabet, size=nsym, p=weights).astype(np.uint8)
    table = np.ones((6400, 80), dtype=np.uint32)
    ctxs = rng.integers(0, 6400, size=200_000)
    np.add.at(table, (ctxs, rng.choice(alphabet, size=ctxs.size, p=weights)), rng.integers(1, 50, size=ctxs.size).astype(np.uint32))
    table[:, alphabet] += 3
    if case == "rare_outside":  # symbols 5 and 70 keep the floor count everywhere, and occur
        at = rng.integers(2, nsym, size=4000)
        sym[at] = rng.choice(np.array([5, 70, 19, 42], dtype=np.uint8), size=at.size)
        sym[100:103] = (5, 5, 70)
    table = table.reshape(-1)
    if case == "plain_kernel":
        monkeypatch.setenv("SCALCE_AC_DECODE_WPB", "0")
    want = O.AcStat(table).encode_stream(sym)
    b = host.Batch(ctx, 100, max_reads=1024, max_text=1 << 20)
    d_sym = torch.from_numpy(sym).to("cuda:0")
    d_tab = torch.from_numpy(table.view(np.int32)).to("cuda:0")
    b.entropy_stream(0, d_tab.data_ptr(), d_sym.data_ptr(), nsym)
    b.finish()
    enc = b.output(host.OUT_QUAL, 0)
    assert len(enc) == len(want) and (enc == want).all(), f"{case}: coder bytes differ from the oracle's"
    out = torch.zeros(nsym, dtype=torch.uint8, device="cuda:0")
    p, nbytes = b.output_ptr(host.OUT_QUAL, 0)
    ctx.ac_decode(table, p, nbytes, nsym, out.data_ptr())
    got = out.cpu().numpy()
    bad = np.flatnonzero(got != sym)
    assert len(bad) == 0, f"{case}: decoded stream differs first at {bad[:5]}: {got[bad[:5]]} vs {sym[bad[:5]]}"



@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["1", "4", "8", "64", "grouped"])
def test_coder_block_that_outgrows_its_buffer_is_coded_again(variant, ctx, monkeypatch):
    """The coder's block buffers are sized from what the table says (ac_prepare), not at the reference's 10 MiB per block
    (arithmetic.cpp:301).  With the estimate forced far too small every block overflows, the shard reports it, and the
    collect step codes it again at the full stride: the same bytes as a run that had the room from the start."""
    from gpu_util import device_bytes
    n, L = 230_000, 100                                   # three blocks
    bases, quals = synth.reads_and_quals(n, L, seed=78)
    fq = synth.fastq_bytes_fast(bases, quals)
    t = device_bytes(fq)
    if variant != "grouped":
        monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", variant)

    def run():
        b = host.Batch(ctx, L, n + 8, len(fq) + 64)
        if variant == "grouped":
            b.front(t.data_ptr(), len(fq))
            host.entropy_begin_group([b])
        else:
            b.compress(t.data_ptr(), len(fq))
        b.finish()
        return b.output(host.OUT_QUAL, 0).copy()
    want = run()
    monkeypatch.setenv("SCALCE_AC_STRIDE_SCALE", "0.5")
    got = run()
    assert len(got) == len(want) and (got == want).all()
    monkeypatch.setenv("SCALCE_AC_STRIDE_SCALE", "0")   # the reference's 10 MiB per block
    got = run()
    assert len(got) == len(want) and (got == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("grouped", [False, True])
def test_frames_on_the_way_out(grouped, ctx):
    """scalce_batch_set_frame_on_demand: the coder's blocks are framed by the kernel that delivers them
    (scalce_batch_qual_window) -- any window of the stream, into device memory and into pinned host memory, is the same
    bytes as the stream framed behind the coder ([u32 size][bytes] per 10 MiB block, arithmetic.cpp:318-363); windows that
    start and end inside size words, inside blocks, on block edges; a caller that asks for SCALCE_OUT_QUAL gets the whole
    stream put together at that moment."""
    import torch
    from gpu_util import device_bytes
    n, L = 330_000, 100                                   # four blocks, the last one short
    bases, quals = synth.reads_and_quals(n, L, seed=77)
    fq = synth.fastq_bytes_fast(bases, quals)
    t = device_bytes(fq)

    def run(on_demand):
        b = host.Batch(ctx, L, n + 8, len(fq) + 64)
        if on_demand:
            b.set_frame_on_demand(True)
        if grouped:
            b.front(t.data_ptr(), len(fq))
            host.entropy_begin_group([b])
        else:
            b.compress(t.data_ptr(), len(fq))
        b.finish()
        return b
    want = run(False).output(host.OUT_QUAL, 0).copy()
    b = run(True)
    total = b.qual_bytes(0)
    assert total == len(want)
    sizes, pos = [], 0
    while pos < total:                                    # the frames, from the expected stream
        sz = int.from_bytes(want[pos:pos + 4].tobytes(), "little")
        sizes.append((pos, sz))
        pos += 4 + sz
    assert len(sizes) == 4
    rng = np.random.default_rng(3)
    wins = [(0, total), (0, 1), (1, 2), (3, 9), (total - 5, 5), (total - 1, 1), (0, 0)]
    for off, sz in sizes:                                 # around every frame edge
        for a in (off - 7, off - 1, off, off + 1, off + 3, off + 4, off + 5):
            for k in (1, 2, 3, 4, 5, 8, 13, 4096 + 3):
                if a >= 0 and a + k <= total:
                    wins.append((a, k))
    for _ in range(40):
        a = int(rng.integers(0, total))
        wins.append((a, int(rng.integers(0, min(total - a, 3 << 20) + 1))))
    dev = torch.empty(total + 64, dtype=torch.uint8, device="cuda:0")
    pin = torch.empty(total + 64, dtype=torch.uint8).pin_memory()
    for a, k in wins:
        for buf in (dev, pin):
            buf.fill_(0xEE)
            torch.cuda.synchronize()
            b.qual_window(0, a, k, buf.data_ptr())
            torch.cuda.synchronize()
            got = buf.cpu().numpy() if buf is dev else buf.numpy()
            assert (got[:k] == want[a:a + k]).all(), (a, k, "device" if buf is dev else "pinned")
            assert (got[k:k + 32] == 0xEE).all(), (a, k, "wrote past the window")
    got = b.output(host.OUT_QUAL, 0)                      # put together on request
    assert len(got) == len(want) and (got == want).all()
    b.qual_window(0, 5, 100, dev.data_ptr())              # ... and windows of the stream that now exists
    torch.cuda.synchronize()
    assert (dev[:100].cpu().numpy() == want[5:105]).all()


@pytest.mark.gpu
def test_rows_coder_repeated_launches(ctx, oracle_trie, monkeypatch):
    """The same two-block shard through the four- and eight-blocks-per-wave coder a dozen times, every result against
    the oracle's bytes.  (The hand-over of the first operand slot between helper and chain waves once raced: a block
    coded wrongly in one launch out of a few.  It showed in test_rows_coder_redo_path[8] when it ran behind
    test_grouped_coder_launch_equals_separate_launches -- keep those two in that order -- and not in this loop on its
    own; repetition is cheap all the same.)"""
    from gpu_util import hip_compress, oracle_streams
    bases, quals = synth.reads_and_quals(150_000, 100, seed=78)
    fq = synth.fastq_bytes_fast(bases, quals)
    ref = oracle_streams(oracle_trie, bases, quals, 33, None)
    want = None
    for rep in range(6):
        for bpw in ("8", "4", "64"):
            monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", bpw)
            b = hip_compress(ctx, fq, 100)
            if want is None:
                want = O.AcStat(b.output(host.OUT_TABLE, 0, np.uint32)).encode_stream(ref["qp"][ref["perm"]].reshape(-1))
            enc = b.output(host.OUT_QUAL, 0)
            assert len(enc) == len(want) and (enc == want).all(), f"launch {rep} with {bpw} blocks per wave differs"


@pytest.mark.gpu
@pytest.mark.parametrize("blocks_per_wave", ["1", "4", "8", "64"])   # (64: the host falls back to eight per wave)
def test_coder_general_step_through_inverted_intervals(blocks_per_wave, ctx, monkeypatch):
    """A context total beyond 2^30 (reachable at 50 M x 100 with binned, low-entropy qualities) makes the reference's
    32-bit coder run through inverted intervals: lo = 1.., hi = 0.. after a step (about 1 % of the steps with this
    table).  The host then selects the all-states step; the rows kernels carry (lo, range) across super-rounds and must
    not clear bit 31 of lo there (they once did: ADVICE r1).  One-, four- and eight-blocks-per-wave kernels against the
    oracle's literal coder, two blocks, the second one short."""
    import torch
    monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", blocks_per_wave)
    rng = np.random.default_rng(9)
    nsym = 10 * 1024 * 1024 + 300_000
    sym = rng.choice(np.array([30, 31, 32], dtype=np.uint8), size=nsym, p=[0.62, 0.03, 0.35]).astype(np.uint8)
    table = np.ones((6400, 80), dtype=np.uint32)
    table[:, 30] = 2_000_000_000
    table[:, 32] = 1_000_000_000          # context totals 3e9 + 78: above 2^30, below 2^32
    table = table.reshape(-1)
    want = O.AcStat(table).encode_stream(sym)
    b = host.Batch(ctx, 100, max_reads=1024, max_text=1 << 20)
    d_sym = torch.from_numpy(sym).to("cuda:0")
    d_tab = torch.from_numpy(table.view(np.int32)).to("cuda:0")
    b.entropy_stream(0, d_tab.data_ptr(), d_sym.data_ptr(), nsym)
    b.finish()
    enc = b.output(host.OUT_QUAL, 0)
    assert len(enc) == len(want), f"{len(enc)} vs {len(want)} bytes"
    bad = np.flatnonzero(enc != want)
    assert len(bad) == 0, f"coder bytes differ from the oracle's first at byte {bad[:3]} of {len(want)}"


@pytest.mark.gpu
@pytest.mark.parametrize("blocks_per_wave", ["1", "8", "64"])
def test_coder_long_underflow_runs_and_carries(blocks_per_wave, ctx, monkeypatch):
    """Symbols chosen with the coder's state in hand (gpu_util.craft_straddle): runs of 1..40 symbols whose interval holds
    the midpoint, so every one of them adds pending underflow bits (arithmetic.cpp:140-146), up to ~260 in a row, and the
    next symbol resolves them all at once -- '1 000..0' or '0 111..1'.  A quarter of the coded words are all ones or all
    zeros.  For the one-block-per-lane coder that is a carry walking back through words it has already stored
    (AclSink::carry_back); for the others the serial emit path.  Two blocks that start with 60 000 such symbols."""
    import torch
    from gpu_util import craft_straddle
    monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", blocks_per_wave)
    rng = np.random.default_rng(4)
    row = np.ones(80, dtype=np.uint32)
    row[8:40] = 1000
    cum = np.concatenate([[0], np.cumsum(row)])
    nsym = 10 * 1024 * 1024 + 150_000
    sym = rng.integers(8, 40, size=nsym).astype(np.uint8)
    for start in (0, 10 * 1024 * 1024):
        part, maxpend = craft_straddle(60_000, cum, int(cum[-1]), rng)
        assert maxpend > 128
        sym[start:start + len(part)] = part
    table = np.tile(row, 6400)
    want = O.AcStat(table).encode_stream(sym)
    words = np.frombuffer(want[4:4 + 36000].tobytes(), dtype=np.uint32)
    assert (words == 0xFFFFFFFF).sum() > 500 and (words == 0).sum() > 500   # the case is what it claims to be
    b = host.Batch(ctx, 100, max_reads=1024, max_text=1 << 20)
    d_sym = torch.from_numpy(sym).to("cuda:0")
    d_tab = torch.from_numpy(table.view(np.int32)).to("cuda:0")
    b.entropy_stream(0, d_tab.data_ptr(), d_sym.data_ptr(), nsym)
    b.finish()
    enc = b.output(host.OUT_QUAL, 0)
    assert len(enc) == len(want), f"{len(enc)} vs {len(want)} bytes"
    bad = np.flatnonzero(enc != want)
    assert len(bad) == 0, f"coder bytes differ from the oracle's first at byte {bad[:3]} of {len(want)}"


@pytest.mark.gpu
def test_lanes_coder_shapes_write_the_same_bytes(ctx, monkeypatch):
    """ac_encode_lanes_k in every shape of its workgroup -- one set of four waves over rows of 48 or 64 lanes, two sets over
    rows of 48, 40 and 32 (eight waves, two per SIMD; kernels_acl.hpp) -- and with every seventh round forced through the
    redo path: 130 blocks (more than one workgroup in every shape, a last workgroup that is partly empty, a last block that
    is short), crafted carries at the start of two of them.  All shapes against the first, and the first block of the
    stream -- which holds the crafted symbols -- against the oracle's bytes."""
    import hashlib
    import torch
    from gpu_util import craft_straddle
    monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", "64")
    rng = np.random.default_rng(11)
    row = np.ones(80, dtype=np.uint32)
    row[8:40] = 1000
    cum = np.concatenate([[0], np.cumsum(row)])
    BLK = 10 * 1024 * 1024
    nsym = 129 * BLK + 77_777
    g = torch.Generator(device="cuda:0"); g.manual_seed(3)
    d_sym = torch.randint(8, 40, (nsym + 64,), dtype=torch.uint8, device="cuda:0", generator=g)
    part, maxpend = craft_straddle(40_000, cum, int(cum[-1]), rng)
    assert maxpend > 64
    for start in (0, 97 * BLK):
        d_sym[start:start + len(part)] = torch.from_numpy(part).to("cuda:0")
    table = np.tile(row, 6400)
    d_tab = torch.from_numpy(table.view(np.int32)).to("cuda:0")
    first = d_sym[:BLK].cpu().numpy()
    want0 = O.AcStat(table).encode_stream(first)
    ref = None
    # (8 .. 28: what a launch that has the chip to itself may pick -- as few blocks per workgroup as all CUs allow)
    for sets, lanes, poison in (("1", "48", "0"), ("1", "64", "0"), ("1", "32", "0"), ("1", "32", "7"), ("1", "40", "7"), ("1", "8", "0"),
                                ("1", "11", "7"), ("1", "28", "0")):
        monkeypatch.setenv("SCALCE_AC_LANES_USED", lanes)
        monkeypatch.setenv("SCALCE_AC_TEST_POISON", poison)
        b = host.Batch(ctx, 100, max_reads=1024, max_text=1 << 20)
        b.entropy_stream(0, d_tab.data_ptr(), d_sym.data_ptr(), nsym)
        b.finish()
        enc = b.output(host.OUT_QUAL, 0)
        if ref is None:
            assert (enc[:len(want0)] == want0).all(), "the first block differs from the oracle's"
            ref = (len(enc), hashlib.sha256(enc.tobytes()).hexdigest())
        else:
            assert (len(enc), hashlib.sha256(enc.tobytes()).hexdigest()) == ref, f"sets {sets}, rows of {lanes} lanes: other bytes"
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nsym", [1, 2, 3, 15, 16, 17, 33, 10 * 1024 * 1024, 10 * 1024 * 1024 + 1, 10 * 1024 * 1024 + 2])
def test_lanes_coder_block_edges(nsym, ctx, monkeypatch):
    """One block per lane: blocks of 1, 2 and 3 symbols (nothing but the raw symbols and the flush), a round that is
    exactly full, a second block of one or two symbols behind a full one."""
    import torch
    monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", "64")
    rng = np.random.default_rng(nsym % 1000)
    sym = np.clip(np.rint(rng.normal(28, 8, size=nsym)), 0, 41).astype(np.uint8)
    table = (rng.integers(1, 200, size=512000)).astype(np.uint32)
    want = O.AcStat(table).encode_stream(sym)
    b = host.Batch(ctx, 100, max_reads=1024, max_text=1 << 20)
    d_sym = torch.from_numpy(np.concatenate([sym, np.zeros(64, np.uint8)])).to("cuda:0")
    d_tab = torch.from_numpy(table.view(np.int32)).to("cuda:0")
    b.entropy_stream(0, d_tab.data_ptr(), d_sym.data_ptr(), nsym)
    b.finish()
    enc = b.output(host.OUT_QUAL, 0)
    assert len(enc) == len(want) and (enc == want).all(), f"{nsym} symbols: {len(enc)} vs {len(want)} bytes"


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["indexed_kernels", "records_longer_than_the_overlap", "short_reads_L16", "tile_edges", "odd_length_L75"])
def test_ingest_paths_agree(case, ctx, oracle_trie, monkeypatch):
    """The ingest stage reads the text once behind the newline count (ingest_tiles2_k: 16 KB tiles + 1 KB overlap in LDS,
    the unpack dealt to the lanes word by word) and
    falls back to the indexed kernels (index_write_k + unpack_tiled_k) for read lengths outside 16..160 or when a record
    does not fit the overlap.  All paths, the fallback trigger, the shortest fused read length, a read length that is not
    a multiple of four and records that straddle tile boundaries in every phase against the oracle."""
    from gpu_util import hip_compress, oracle_streams
    rng = np.random.default_rng(17)
    L, n = 100, 40000
    if case.startswith("short_reads_L16"):
        L, n = 16, 150000
    if case == "odd_length_L75":
        L = 75
    bases, quals = synth.reads_and_quals(n, L, seed=55, n_frac=0.004, dup_frac=0.1)
    if case == "indexed_kernels":
        monkeypatch.setenv("SCALCE_INGEST_INDEXED", "1")
    if case.startswith("records_longer_than_the_overlap"):   # names of up to 255 characters, repeated on the '+' line
        recs = []
        for i in range(n):
            nm = b"r%d_" % i + b"x" * int(rng.integers(0, 250 - 8))
            recs.append(b"@" + nm + b"\n" + bases[i].tobytes() + b"\n+" + nm + b"\n" + quals[i].tobytes() + b"\n")
        fq = b"".join(recs)
    elif case.startswith("tile_edges"):             # name lengths cycle so that record starts sweep through every tile phase
        recs = [b"@" + b"n" * (1 + (i * 7) % 61) + b" c\n" + bases[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n" for i in range(n)]
        fq = b"".join(recs)
    else:
        fq = synth.fastq_bytes_fast(bases, quals)
    b = hip_compress(ctx, fq, L)
    ref = oracle_streams(oracle_trie, bases, quals, 33, None)
    tok = b.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)
    assert (tok[:, 0] == ref["pat"]).all() and (tok[:, 1] == ref["end"]).all()
    assert (b.output(host.OUT_PERM, 0, np.uint32) == ref["perm"]).all()
    assert (b.output(host.OUT_QINPUT, 0).reshape(n, L) == ref["qp"]).all()
    assert (b.output(host.OUT_FREQ4, 0, np.uint64) + 1 == ref["f4"]).all()
    # names: [u8 n][chars up to the first space] per record, in emission order
    names = b.output(host.OUT_NAMES, 0)
    lines = fq.split(b"\n")
    want = bytearray()
    for k in ref["perm"]:
        nm = lines[4 * int(k)][1:].split(b" ")[0]
        want.append(len(nm))
        want += nm
    assert names.tobytes() == bytes(want)
