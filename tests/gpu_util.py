"""Helpers for the -m gpu parity tests: run the HIP path through the C ABI, fetch every stream."""
import numpy as np

import oraclelib as O
from scalce_amd import host, synth


def device_bytes(data):
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    assert t.data_ptr() % 16 == 0
    return t


def hip_compress(ctx, fastq1, L, fastq2=None, L2=0, **kw):
    import torch
    t1 = device_bytes(fastq1)
    t2 = device_bytes(fastq2) if fastq2 is not None else None
    nrec = fastq1.count(b"\n") // 4
    b = host.Batch(ctx, L, max_reads=nrec + 8, max_text=max(len(fastq1), len(fastq2 or b"")) + 64,
                   paired=fastq2 is not None, read_len2=L2, **kw)
    b.compress(t1.data_ptr(), len(fastq1), t2.data_ptr() if t2 is not None else None, len(fastq2 or b""))
    b.finish()
    torch.cuda.synchronize()
    b._keep = (t1, t2)
    return b


def oracle_streams(trie, bases, quals, qoff=33, qvals=None, chunk=None, no_ac=False):
    """Everything the oracle says about one mate-1 shard."""
    qvals = np.arange(128) if qvals is None else qvals
    pat, end = trie.tokenize(bases)
    perm = trie.order(bases, pat, end, chunk)
    qp, f4 = O.quality_stream(quals, bases, qoff, qvals, no_ac=no_ac)
    return dict(pat=pat, end=end, perm=perm, qp=qp, f4=f4)


def craft_straddle(n, cum, total, rng, first2=(9, 12)):
    """n symbols for a context-free table (cum[0..80] cumulative counts) chosen while following the reference coder's
    state (arithmetic.cpp:122-152): runs of 1..40 symbols whose interval holds the midpoint -- each adds pending
    underflow bits -- between stretches of random symbols.  Returns (symbols, longest pending run in bits)."""
    out = [first2[0], first2[1]]
    lo, hi = 0, 0xFFFFFFFF
    used = [s for s in range(80) if cum[s + 1] > cum[s] + 1]
    run_left, free, pend, maxpend = 0, 0, 0, 0
    while len(out) < n:
        if run_left == 0 and free == 0:
            run_left, free = int(rng.integers(1, 41)), int(rng.integers(1, 12))
        width = ((hi - lo) & 0xFFFFFFFF) + 1
        pick = None
        if free == 0:
            for s in used:
                nl = lo + width * int(cum[s]) // total
                nh = lo + width * int(cum[s + 1]) // total - 1
                if 0x40000000 <= nl < 0x80000000 <= nh < 0xC0000000:
                    pick = s
                    break
            run_left -= 1
        if pick is None:
            pick = int(rng.choice(used))
            free = max(0, free - 1)
        nh = (lo + width * int(cum[pick + 1]) // total - 1) & 0xFFFFFFFF
        nl = (lo + width * int(cum[pick]) // total) & 0xFFFFFFFF
        while True:
            if (nh & 0x80000000) == (nl & 0x80000000):
                pend = 0
            elif not (nh & 0x40000000) and (nl & 0x40000000):
                nl &= 0x3FFFFFFF
                nh |= 0x40000000
                pend += 1
                maxpend = max(maxpend, pend)
            else:
                break
            nl = (nl << 1) & 0xFFFFFFFF
            nh = ((nh << 1) | 1) & 0xFFFFFFFF
        lo, hi = nl, nh
        out.append(pick)
    return np.array(out[:n], dtype=np.uint8), maxpend
