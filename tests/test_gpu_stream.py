"""-m gpu: a run ingested piece by piece (scalce_batch_append: the text is streamed, the derived rows stay in HBM)
gives byte for byte what the same input gives as one resident shard -- tokens incl. the tie-break across pieces,
the trigram counters across piece boundaries, order, names, records, coded qualities -- and both equal the oracle."""
import os

import numpy as np
import pytest

import oraclelib as O
from scalce_amd import host, synth

pytestmark = pytest.mark.gpu
OUTS = [(host.OUT_TOKENS, 0), (host.OUT_PERM, 0), (host.OUT_QINPUT, 0), (host.OUT_FREQ4, 0), (host.OUT_BUCKET_COUNTS, 0),
        (host.OUT_READS, 0), (host.OUT_NAMES, 0), (host.OUT_TABLE, 0), (host.OUT_QSTREAM, 0), (host.OUT_QUAL, 0)]


def feed_in_pieces(b, texts, piece, rng=None):
    """Streams `texts` (one byte string per mate) through Batch.append the way a file reader would: raw chunks of
    `piece` bytes cut anywhere, the unconsumed tail of a piece in front of the next chunk."""
    from gpu_util import device_bytes
    nm = len(texts)
    pos = [0] * nm
    tail = [b""] * nm
    rounds = 0
    while True:
        bufs, final = [], True
        for m in range(nm):
            take = piece if rng is None else int(rng.integers(max(1, piece // 2), piece + 1))
            room = max(0, take - len(tail[m]))
            chunk = texts[m][pos[m]:pos[m] + room]
            pos[m] += len(chunk)
            bufs.append(tail[m] + chunk)
            final = final and pos[m] >= len(texts[m])
        dev = [device_bytes(x) if len(x) else None for x in bufs]
        used = b.append(dev[0].data_ptr() if dev[0] is not None else None, len(bufs[0]),
                        dev[1].data_ptr() if nm == 2 and dev[1] is not None else None, len(bufs[1]) if nm == 2 else 0, final=final)
        rounds += 1
        for m in range(nm):
            tail[m] = bufs[m][used[m]:]
        if final:
            assert all(len(t) == 0 for t in tail)
            return rounds
        assert rounds < 100000


@pytest.mark.parametrize("case", ["se100", "se36_ties", "pe150_lossy", "long_names", "noac_nonames"])
def test_pieces_equal_one_shard(case, patterns_blob):
    from gpu_util import device_bytes
    rng = np.random.default_rng(3)
    kw, paired, L, n = {}, False, 100, 40000
    ptxt = None
    if case == "se36_ties":  # every position an equal-length tie: the cumulative counts across pieces decide nearly every read
        import itertools
        ptxt = ("\n".join("".join(x) for x in itertools.product("ACGT", repeat=4)) + "\n").encode()
        L, n = 36, 30000
    if case.startswith("pe150"):
        paired, L, n = True, 150, 20000
    ctx = host.Context(0, patterns_text=ptxt) if ptxt else host.Context(0, patterns_bin=patterns_blob)
    bases, quals = synth.reads_and_quals(n, L, seed=101, dup_frac=0.15, n_frac=0.003)
    if case == "long_names":  # names of 1..60 characters, some with a comment behind a space: cells and the long-name store
        recs = []
        for i in range(n):
            nm = ("r%d" % i) + "x" * int(rng.integers(0, 58))
            if i % 7 == 0:
                nm += " comment %d" % i
            recs.append(b"@" + nm.encode() + b"\n" + bases[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")
        fq1 = b"".join(recs)
    else:
        fq1 = synth.fastq_bytes_fast(bases, quals, prefix="p." if paired else "s.", suffix="/1" if paired else "")
    texts = [fq1]
    qm = None
    if paired:
        bases2, quals2 = synth.reads_and_quals(n, L, seed=102, n_frac=0.003)
        texts.append(synth.fastq_bytes_fast(bases2, quals2, prefix="p.", suffix="/2"))
        from scalce_amd import format as fmt
        qm = [fmt.sample_qmap(texts[0], lossy=30)[:2], fmt.sample_qmap(texts[1], lossy=30)[:2]]
    if case == "noac_nonames":
        kw = dict(no_ac=True, use_names=False)
    whole = host.Batch(ctx, L, n + 8, max(len(t) for t in texts) + 64, paired=paired, read_len2=L, qmap=qm, **kw)
    dev = [device_bytes(t) for t in texts]
    whole.compress(dev[0].data_ptr(), len(texts[0]), dev[1].data_ptr() if paired else None, len(texts[1]) if paired else 0)
    whole.finish()
    piece = len(texts[0]) // 7 + 13
    # capacity below the run: the row arrays have to grow while the run is ingested
    b = host.Batch(ctx, L, n // 3, piece + 64, paired=paired, read_len2=L, qmap=qm, **kw)
    rounds = feed_in_pieces(b, texts, piece, rng)
    assert rounds >= 7 and b.n_reads == n
    b.order(); b.emit(); b.entropy(); b.finish()
    outs = OUTS + ([(host.OUT_READS, 1), (host.OUT_QUAL, 1), (host.OUT_FREQ4, 1)] if paired else [])
    for which, m in outs:
        if kw.get("no_ac") and which in (host.OUT_TABLE,):
            continue
        x, y = whole.output(which, m), b.output(which, m)
        assert len(x) == len(y), (which, m, len(x), len(y))
        bad = np.flatnonzero(x != y)
        assert len(bad) == 0, f"output {which} mate {m}: differs first at byte {bad[:4]} of {len(x)}"
    # and the one-piece result is the oracle's (tokens and order; the rest is covered by test_gpu_parity)
    trie = O.Trie(text=ptxt) if ptxt else O.Trie(blob=patterns_blob)
    pat, end = trie.tokenize(bases)
    tok = b.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)
    assert (tok[:, 0] == pat).all() and (tok[:, 1] == end).all()
    assert (b.output(host.OUT_PERM, 0, np.uint32) == trie.order(bases, pat, end)).all()


def test_append_with_spill_chunks_and_reuse(patterns_blob):
    """-B cuts chunks on the run-wide record sizes, wherever the pieces were cut; a batch can be reset and reused."""
    from gpu_util import device_bytes
    ctx = host.Context(0, patterns_bin=patterns_blob)
    n, L = 30000, 100
    for seed in (5, 6):
        bases, quals = synth.reads_and_quals(n, L, seed=seed, dup_frac=0.2)
        fq = synth.fastq_bytes_fast(bases, quals)
        whole = host.Batch(ctx, L, n + 8, len(fq) + 64, bucket_set_size=900_000)
        t = device_bytes(fq)
        whole.compress(t.data_ptr(), len(fq))
        whole.finish()
        assert whole.stats()["chunks"] > 3
        if seed == 5:
            b = host.Batch(ctx, L, n + 8, len(fq) // 5 + 64, bucket_set_size=900_000)
        else:
            b.reset()
        feed_in_pieces(b, [fq], len(fq) // 5)
        b.order(); b.emit(); b.entropy(); b.finish()
        assert b.stats()["chunks"] == whole.stats()["chunks"]
        for which in (host.OUT_PERM, host.OUT_READS, host.OUT_NAMES, host.OUT_QUAL):
            assert (whole.output(which, 0) == b.output(which, 0)).all(), which


def test_append_errors(patterns_blob):
    from gpu_util import device_bytes
    ctx = host.Context(0, patterns_bin=patterns_blob)
    bases, quals = synth.reads_and_quals(100, 50, seed=1)
    fq = synth.fastq_bytes_fast(bases, quals)
    b = host.Batch(ctx, 50, 200, len(fq) + 64)
    t = device_bytes(fq[:-10])  # the last record is cut short
    used = b.append(t.data_ptr(), len(fq) - 10)
    assert b.n_reads == 99 and 0 < used[0] < len(fq) - 10 and fq[used[0] - 1:used[0]] == b"\n"
    with pytest.raises(host.ScalceError):
        rest = fq[used[0]:len(fq) - 10]
        tr = device_bytes(rest)
        b.append(tr.data_ptr(), len(rest), final=True)


@pytest.mark.parametrize("paired", [False, True])
def test_windowed_coder_equals_one_launch(paired, patterns_blob, monkeypatch):
    """Runs with more than 2048 blocks are coded window by window into bounded buffers (entropy_windowed); with the window
    shrunk to one block per mate the same path runs here: three windows, the framed stream grown twice."""
    from gpu_util import device_bytes
    ctx = host.Context(0, patterns_bin=patterns_blob)
    n, L = 230_000, 100   # three blocks per mate, the last one short
    texts = [synth.fastq_bytes_fast(*synth.reads_and_quals(n, L, seed=71 + m)) for m in range(2 if paired else 1)]
    dev = [device_bytes(t) for t in texts]
    outs = []
    for windowed in (False, True):
        if windowed:
            monkeypatch.setenv("SCALCE_AC_WINDOW_BLOCKS", "1")
        b = host.Batch(ctx, L, n + 8, max(len(t) for t in texts) + 64, paired=paired, read_len2=L)
        b.compress(dev[0].data_ptr(), len(texts[0]), dev[1].data_ptr() if paired else None, len(texts[1]) if paired else 0)
        b.finish()
        outs.append([b.output(host.OUT_QUAL, m).copy() for m in range(len(texts))])
    for m in range(len(texts)):
        assert len(outs[0][m]) == len(outs[1][m]) and (outs[0][m] == outs[1][m]).all(), f"mate {m + 1}"


def _params(L, paired=False, **kw):
    import ctypes as C
    p = host.Params()
    host.lib().scalce_params_default(C.byref(p))
    p.read_len[0] = L
    p.read_len[1] = L if paired else 0
    p.paired = int(paired)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@pytest.mark.parametrize("case", ["dribble", "paired_hint_too_small", "read_error", "mate_runs_dry"])
def test_stream_compress_entry_point(case, patterns_blob):
    """scalce_stream_compress through its C entry with Python callbacks as the readers: a source that delivers a few
    hundred bytes per call, row arrays that have to grow past the hint, a source that fails, a mate that ends early."""
    from gpu_util import device_bytes
    ctx = host.Context(0, patterns_bin=patterns_blob)
    paired = case in ("paired_hint_too_small", "mate_runs_dry")
    n, L = 30000, 100
    texts = [synth.fastq_bytes_fast(*synth.reads_and_quals(n, L, seed=211 + m, dup_frac=0.1), prefix="p.", suffix="/%d" % (m + 1))
             for m in range(2 if paired else 1)]
    if case == "mate_runs_dry":
        texts[1] = texts[1][:len(texts[1]) // 2]
        texts[1] = texts[1][:texts[1].rfind(b"\n@p.") + 1]

    def reader(text, step, fail_at=None):
        pos = [0]

        def rd(cap):
            if fail_at is not None and pos[0] >= fail_at:
                raise IOError("boom")
            k = min(cap, step, len(text) - pos[0])
            out = text[pos[0]:pos[0] + k]
            pos[0] += k
            return out
        return rd

    step = 700 if case == "dribble" else 1 << 20
    r1 = reader(texts[0], step, fail_at=len(texts[0]) // 2 if case == "read_error" else None)
    r2 = reader(texts[1], step) if paired else None
    p = _params(L, paired, bucket_set_size=0)
    if case in ("read_error", "mate_runs_dry"):
        with pytest.raises(host.ScalceError):
            host.stream_compress(ctx, p, r1, r2, piece_bytes=300_000)
        return
    b, st = host.stream_compress(ctx, p, r1, r2, piece_bytes=300_000, reads_hint=1000 if paired else 0)
    assert b.n_reads == n and st.reads == n and st.rounds > 5
    whole = host.Batch(ctx, L, n + 8, max(len(t) for t in texts) + 64, paired=paired, read_len2=L)
    dev = [device_bytes(t) for t in texts]
    whole.compress(dev[0].data_ptr(), len(texts[0]), dev[1].data_ptr() if paired else None, len(texts[1]) if paired else 0)
    whole.finish()
    for which, m in [(host.OUT_READS, 0), (host.OUT_NAMES, 0), (host.OUT_QUAL, 0)] + ([(host.OUT_READS, 1), (host.OUT_QUAL, 1)] if paired else []):
        x, y = whole.output(which, m), b.output(which, m)
        assert len(x) == len(y) and (x == y).all(), (which, m)


def test_reused_batch_virtual_frames_do_not_leak_into_the_windowed_path(patterns_blob, monkeypatch):
    """a batch with frames on demand that coded a small shard through a grouped launch keeps the LAYOUT of that
    shard's frames (frame_virtual / frame_off_host); if its next shard goes window by window (entropy_windowed writes the
    framed stream itself) scalce_batch_qual_window / SCALCE_OUT_QUAL must not serve the stale layout."""
    from gpu_util import device_bytes
    ctx = host.Context(0, patterns_bin=patterns_blob)
    L = 100
    small = synth.fastq_bytes_fast(*synth.reads_and_quals(30_000, L, seed=81))
    big = synth.fastq_bytes_fast(*synth.reads_and_quals(230_000, L, seed=82))
    t_small, t_big = device_bytes(small), device_bytes(big)
    fresh = host.Batch(ctx, L, 230_008, len(big) + 64)
    fresh.compress(t_big.data_ptr(), len(big))
    fresh.finish()
    want = fresh.output(host.OUT_QUAL, 0).copy()
    b = host.Batch(ctx, L, 230_008, len(big) + 64)
    b.set_frame_on_demand(True)
    b.front(t_small.data_ptr(), len(small))
    host.entropy_begin_group([b])          # virtual frames of the small shard
    b.finish()
    assert len(b.output(host.OUT_QUAL, 0)) > 0
    monkeypatch.setenv("SCALCE_AC_WINDOW_BLOCKS", "1")
    b.compress(t_big.data_ptr(), len(big))   # three windows of one block: the framed stream is written by the windows
    b.finish()
    got = b.output(host.OUT_QUAL, 0)
    assert len(got) == len(want) and (got == want).all()
    n = b.qual_bytes(0)
    assert n == len(want)
