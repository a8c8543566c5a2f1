"""-m gpu: the tokenizer at the sizes and in the corners the reference admits but no other test reaches -- a core table of
a million cores of 12-32 bases (/root/reference/reads.cpp:336,353-358 allow 5 M x 32), and inputs whose tie-break is as
sequential as the reference's own loop (reads.cpp:413-429 with the cumulative bin_size of :246)."""
import itertools

import numpy as np
import pytest

import bigtable
import oraclelib as O
from gpu_util import device_bytes
from scalce_amd import host, synth

pytestmark = pytest.mark.gpu


def run_tokens(ctx, bases, L):
    quals = np.full(bases.shape, ord("I"), dtype=np.uint8)
    fq = synth.fastq_bytes_fast(bases, quals)
    t = device_bytes(fq)
    b = host.Batch(ctx, L, len(bases) + 8, len(fq) + 64)
    b.compress(t.data_ptr(), len(fq))
    b.finish()
    return b


def test_million_core_table():
    """1 M cores of 12..32 bases = 9.9 M automaton states (the k-mer tables in LDS only cover states of depth <= 7: four of
    five transitions go back to 16-byte rows in L2 / HBM): tokens and order against the oracle's trie walk."""
    blob, vals = bigtable.build()
    ctx = host.Context(0, patterns_bin=blob)
    assert ctx.n_patterns == 1_000_000 and ctx.n_states > 9_000_000
    n, L = 300_000, 100
    bases = bigtable.reads_with_cores(n, L, vals)
    b = run_tokens(ctx, bases, L)
    trie = O.Trie(blob=blob)
    pat, end = trie.tokenize(bases)
    assert (pat >= 0).mean() > 0.8
    tok = b.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)
    assert (tok[:, 0] == pat).all() and (tok[:, 1] == end).all()
    assert (b.output(host.OUT_PERM, 0, np.uint32) == trie.order(bases, pat, end)).all()


def stress_inputs(kind, n, L, rng):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    if kind == "pileup":           # reads sampled at ~50x from a short sequence: the same few cores win again and again
        g = acgt[rng.integers(0, 4, size=n * L // 50 + L)]
        at = rng.integers(0, len(g) - L, size=n)
        return g[at[:, None] + np.arange(L)[None, :]]
    if kind == "two_cores":        # two equally long cores alternate in every read: every read is a tie between the two
        a, c = b"ACGTTGCAAC", b"TTGACCAGTA"
        unit = np.frombuffer(a + c, dtype=np.uint8)
        rows = np.tile(unit, (n, L // len(unit) + 1))[:, :L].copy()
        flip = rng.random(n) < 0.5          # which of the two comes first differs from read to read
        rows[flip] = np.tile(np.frombuffer(c + a, dtype=np.uint8), (int(flip.sum()), L // len(unit) + 1))[:, :L]
        return rows
    return acgt[rng.integers(0, 4, size=(n, L))]   # "fourmers": random reads against the table of all 4-mers


@pytest.mark.parametrize("kind", ["pileup", "two_cores", "fourmers"])
@pytest.mark.parametrize("window", ["0", "700", "700:two_launches", "40000", "40000:two_launches"])
def test_tie_break_windows(kind, window, monkeypatch):
    """The tie-break in windows of W tie reads (tie_window_fused_k, one launch per sweep; two_launches: tie_window_sweep_k +
    tie_window_tail_k; SCALCE_TIE_WINDOW=0: the global sweeps): windows much smaller than the input, so that hundreds of
    them hand their counts on, give the oracle's tokens on the inputs of test_tie_break_stress."""
    rng = np.random.default_rng(23)
    n, L = 60_000, 100
    text = {"fourmers": "\n".join("".join(x) for x in itertools.product("ACGT", repeat=4)) + "\n",
            "two_cores": "ACGTTGCAAC\nTTGACCAGTA\nGGGGGGGGGG\n"}.get(kind)
    monkeypatch.setenv("SCALCE_TIE_WINDOW", window)
    blob = open(bigtable.__file__.replace("bigtable.py", "golden/patterns.bin"), "rb").read()
    ctx = host.Context(0, patterns_text=text.encode()) if text else host.Context(0, patterns_bin=blob)
    trie = O.Trie(text=text.encode()) if text else O.Trie(blob=blob)
    bases = stress_inputs(kind, n, L, rng)
    b = run_tokens(ctx, bases, L)
    st = b.stats()
    pat, end = trie.tokenize(bases)
    tok = b.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)
    assert (tok[:, 0] == pat).all() and (tok[:, 1] == end).all(), f"{kind} window {window}: tokens differ ({st})"
    assert (b.output(host.OUT_PERM, 0, np.uint32) == trie.order(bases, pat, end)).all()
    print(f"{kind} window={window}: {st}")


@pytest.mark.parametrize("kind", ["pileup", "two_cores", "fourmers"])
@pytest.mark.parametrize("fallback", [False, True])
def test_tie_break_stress(kind, fallback, monkeypatch):
    """Inputs on which the parallel sweeps are slow to settle, with and without the bounded way out (after
    SCALCE_TIE_MAX_SWEEPS sweeps one wavefront decides the tie reads in input order, tie_sequential_k): the same tokens as
    the oracle's sequential loop either way, and the number of sweeps reported."""
    rng = np.random.default_rng(17)
    n, L = 200_000, 100
    if kind == "fourmers":
        text = "\n".join("".join(x) for x in itertools.product("ACGT", repeat=4)) + "\n"
    elif kind == "two_cores":
        text = "ACGTTGCAAC\nTTGACCAGTA\nGGGGGGGGGG\n"
    else:
        text = None
    if fallback:
        monkeypatch.setenv("SCALCE_TIE_MAX_SWEEPS", "8")
    blob = open(bigtable.__file__.replace("bigtable.py", "golden/patterns.bin"), "rb").read()
    ctx = host.Context(0, patterns_text=text.encode()) if text else host.Context(0, patterns_bin=blob)
    trie = O.Trie(text=text.encode()) if text else O.Trie(blob=blob)
    bases = stress_inputs(kind, n, L, rng)
    b = run_tokens(ctx, bases, L)
    st = b.stats()
    pat, end = trie.tokenize(bases)
    tok = b.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)
    assert (tok[:, 0] == pat).all() and (tok[:, 1] == end).all(), f"{kind}: tokens differ ({st})"
    assert (b.output(host.OUT_PERM, 0, np.uint32) == trie.order(bases, pat, end)).all()
    print(f"{kind} fallback={fallback}: {st}")
    if fallback and st["tie_reads"] > 1000:
        assert st["tie_fallback"] == 1 or st["jacobi_iters"] <= 8
