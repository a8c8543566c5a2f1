"""Seeded synthetic FASTQ for tests and bench.py.

names ``@s.<i>`` (paired: ``@p.<i>/1`` / ``/2``), bases iid uniform ACGT, qualities
``clip(round(N(30, 8)), 2, 40) + 33``, bare ``+`` line, numpy ``default_rng(20261003)``.
This is data generation only -- it is not part of the hot path.
"""
import numpy as np

SEED = 20261003
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def reads_and_quals(n, length, seed=SEED, n_frac=0.0, dup_frac=0.0):
    """Return (bases[n, length] uint8 ASCII, quals[n, length] uint8 ASCII)."""
    rng = np.random.default_rng(seed)
    bases = _ACGT[rng.integers(0, 4, size=(n, length), dtype=np.uint8)]
    q = np.clip(np.rint(rng.normal(30.0, 8.0, size=(n, length))), 2, 40).astype(np.uint8) + 33
    if dup_frac > 0 and n > 1:  # duplicated reads exercise the stable in-bucket order
        k = int(n * dup_frac)
        src = rng.integers(0, n, size=k)
        dst = rng.integers(0, n, size=k)
        bases[dst] = bases[src]
    if n_frac > 0:
        mask = rng.random(size=(n, length)) < n_frac
        bases[mask] = ord("N")
    return bases, q


def fastq_bytes(bases, quals, prefix="s.", suffix=""):
    """Assemble FASTQ text (bytes) from base/quality matrices."""
    n, _ = bases.shape
    out = []
    for i in range(n):
        out.append(b"@" + prefix.encode() + str(i).encode() + suffix.encode() + b"\n")
        out.append(bases[i].tobytes() + b"\n+\n")
        out.append(quals[i].tobytes() + b"\n")
    return b"".join(out)


def fastq_bytes_fast(bases, quals, prefix="s.", suffix=""):
    """Vectorised variant of :func:`fastq_bytes` for millions of reads."""
    n, L = bases.shape
    names = np.char.add(np.char.add("@" + prefix, np.arange(n).astype(str)), suffix + "\n").astype("S")
    nl = np.char.str_len(names).astype(np.int64)
    rec = nl + 2 * (L + 1) + 2
    off = np.concatenate([[0], np.cumsum(rec)])
    buf = np.empty(int(off[-1]), dtype=np.uint8)
    w = names.dtype.itemsize
    nm = np.frombuffer(names.tobytes(), dtype=np.uint8).reshape(n, w)
    for k in range(w):
        sel = nl > k
        buf[off[:-1][sel] + k] = nm[sel, k]
    seq0 = off[:-1] + nl
    idx = seq0[:, None] + np.arange(L)[None, :]
    buf[idx] = bases
    buf[seq0 + L] = 10
    buf[seq0 + L + 1] = ord("+")
    buf[seq0 + L + 2] = 10
    buf[idx + L + 3] = quals
    buf[seq0 + 2 * L + 3] = 10
    return buf.tobytes()


def write_fastq(path, n, length, seed=SEED, paired_suffix=None, **kw):
    bases, quals = reads_and_quals(n, length, seed=seed, **kw)
    if paired_suffix is None:
        data = fastq_bytes_fast(bases, quals)
    else:
        data = fastq_bytes_fast(bases, quals, prefix="p.", suffix=paired_suffix)
    with open(path, "wb") as f:
        f.write(data)
    return bases, quals
