"""A core table of realistic size for tests and tools: up to 5 M cores of up to 32 bases are what the reference's loader
admits (/root/reference/reads.cpp:336,353-358); the table every other test uses has 15 600 cores of 8-12 bases."""
import struct

import numpy as np

GROUPS_1M = ((12, 200_000), (14, 150_000), (16, 150_000), (18, 100_000), (20, 100_000), (24, 100_000), (28, 100_000), (32, 100_000))


def build(groups=GROUPS_1M, seed=99):
    """-> (patterns.bin blob, list of (length, values uint64 sorted ascending))"""
    rng = np.random.default_rng(seed)
    out = bytearray()
    vals = []
    for ln, cnt in groups:
        hi = (1 << (2 * ln)) - 1
        v = np.unique(rng.integers(0, hi, size=int(cnt * 1.02) + 16, dtype=np.uint64, endpoint=True))
        v = np.sort(rng.permutation(v)[:cnt])
        assert len(v) == cnt
        nb = (ln + 3) // 4
        out += struct.pack("<hi", ln, cnt)
        out += v.astype("<u8").view(np.uint8).reshape(-1, 8)[:, :nb].tobytes()
        vals.append((ln, v))
    return bytes(out), vals


def bases_of(ln, v):
    """(len(v), ln) uint8 ASCII: base j = (x >> 2 (ln - 1 - j)) & 3 (reads.cpp:346-364)"""
    shifts = (2 * (ln - 1 - np.arange(ln))).astype(np.uint64)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[((v[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.int64)]


def reads_with_cores(n, L, vals, seed=5, planted=0.7, n_frac=0.002):
    """n x L random reads; `planted` of them carry a core of the table at a random place (others may by chance)"""
    rng = np.random.default_rng(seed)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    which = rng.random(n) < planted
    grp = rng.integers(0, len(vals), size=n)
    for g, (ln, v) in enumerate(vals):
        rows = np.flatnonzero(which & (grp == g))
        if not len(rows):
            continue
        pats = bases_of(ln, v[rng.integers(0, len(v), size=len(rows))])
        at = rng.integers(0, L - ln + 1, size=len(rows))
        idx = at[:, None] + np.arange(ln)[None, :]
        bases[rows[:, None], idx] = pats
    if n_frac:
        bases[rng.random((n, L)) < n_frac] = ord("N")
    return bases
