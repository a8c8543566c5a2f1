"""Host-side final writer: wraps the device streams of a Batch into .scalce{n,r,q} files.

Layout follows combine_and_compress_with_split (/root/reference/compress.cpp:263-343): every file
starts with the magic "scalce22"; .scalcer adds int32 no_ac and int32 read length; .scalceq adds
int64 phred offset (mate 1's for both mates) and, unless -A, the 512000 x u32 table and the u64
symbol count; .scalcen adds the names flag byte and, under -n, int64 0 plus the library name.
"""
import gzip
import struct

import numpy as np

from . import host

MAGIC = b"scalce22"


def write_archive(prefix, batch, phred_offset, library=None, gz=False):
    """Write PREFIX_<m>.scalce{n,r,q} for the shard held by `batch` (already compressed + finished)."""
    p = batch.params
    nm = 2 if p.paired else 1
    n = batch.n_reads
    opener = (lambda path: gzip.open(path, "wb", compresslevel=6)) if gz else (lambda path: open(path, "wb"))
    names = batch.output(host.OUT_NAMES, 0).tobytes() if p.use_names else b""
    for m in range(nm):
        L = p.read_len[m]
        with opener(f"{prefix}_{m + 1}.scalcer") as f:
            f.write(MAGIC + struct.pack("<ii", p.no_ac, L))
            f.write(batch.output(host.OUT_READS, m).tobytes())
        qopen = opener if p.no_ac else (lambda path: open(path, "wb"))  # AC output is never containerised (:249)
        with qopen(f"{prefix}_{m + 1}.scalceq") as f:
            f.write(MAGIC + struct.pack("<q", phred_offset))
            if not p.no_ac:
                f.write(batch.output(host.OUT_TABLE, m).tobytes())
                f.write(struct.pack("<Q", n * L))
            f.write(batch.output(host.OUT_QUAL, m).tobytes())
        with opener(f"{prefix}_{m + 1}.scalcen") as f:
            f.write(MAGIC + struct.pack("<B", 1 if p.use_names else 0))
            if p.use_names:
                f.write(names)  # mate 2's file repeats mate 1's names (compress.cpp:450-454)
            else:
                f.write(struct.pack("<q", 0) + (library or "").encode())


def sample_qmap(fastq_bytes, sample=100000, lossy=0):
    """Sampling half of quality_mapping_init (qualities.cpp:64-97): first `sample` records."""
    stat = np.zeros(128, dtype=np.int64)
    pos, L = 0, 0
    data = fastq_bytes
    for _ in range(sample):
        e = pos
        for _k in range(3):
            e = data.find(b"\n", e) + 1
            if e == 0:
                break
        if e == 0:
            break
        q_end = data.find(b"\n", e)
        if q_end < 0:
            break
        q = np.frombuffer(data, dtype=np.uint8, count=q_end - e, offset=e)
        stat += np.bincount(q & 127, minlength=128)
        L = q_end - e
        pos = q_end + 1
    off, vals = host.qmap_init(stat, lossy)
    return off, vals, L
