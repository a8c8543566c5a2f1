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
