"""Self-checks of a timed run (bench.py): does the shard that was just compressed come back?

Host-side plumbing around the C ABI, torch only as device memory: the archive streams of a finished Batch are decoded
on the device (scalce_ac_decode, scalce_fastq_records: the inverse path, /root/reference/decompress.cpp:240-366) and
the rebuilt FASTQ text is compared with the input through an order-independent digest of its records -- the archive
holds the records in bucket order, so the comparison is one of multisets, as in tools/fastq_digest.c.
"""
import numpy as np

from . import host

_P = 0x9E3779B97F4A7C15          # odd: invertible modulo 2^64
_PINV = pow(_P, -1, 1 << 64)
_CHUNK = 1 << 27


def _i64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def _mix(h, torch):
    """64-bit finaliser on int64 tensors (shifts are arithmetic in torch: mask the sign extension away)."""
    h = h ^ ((h >> 31) & 0x1FFFFFFFF)
    h = h * _i64(0x7FB5D329728EA185)
    h = h ^ ((h >> 27) & 0x1FFFFFFFFF)
    h = h * _i64(0x81DADEF4BC2DD44D)
    h = h ^ ((h >> 33) & 0x7FFFFFFF)
    return h


def record_digest(text):
    """(records, sum of a 64-bit hash per record, sum of a second hash) of FASTQ text held in a uint8 device tensor.
    A record is four lines; its hash does not depend on where it stands, so two texts with the same records in any order
    give the same triple.  Everything modulo 2^64 (int64 wrap-around)."""
    import torch
    dev = text.device
    n = text.numel()
    pw = torch.cumprod(torch.full((_CHUNK,), _i64(_P), dtype=torch.int64, device=dev), 0)        # P^(i+1)
    pinv = torch.cumprod(torch.full((_CHUNK,), _i64(_PINV), dtype=torch.int64, device=dev), 0)   # P^-(i+1)
    count, s1, s2 = 0, 0, 0
    pos = 0
    while pos < n:
        end = min(n, pos + _CHUNK)
        seg = text[pos:end]
        nl = (seg == 10).nonzero().flatten()
        k = nl.numel() // 4 * 4
        if k == 0:
            raise ValueError("no complete FASTQ record in %d bytes at offset %d" % (end - pos, pos))
        rec_end = nl[3:k:4]                       # index of each record's last newline
        used = int(rec_end[-1]) + 1
        if end == n and (k != nl.numel() or used != end - pos):
            raise ValueError("text does not end on a record boundary")
        seg = seg[:used].to(torch.int64)
        c = torch.cumsum(seg * pw[:used], 0)      # c[i] = sum_{j <= i} b_j P^(j+1)
        starts = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), rec_end[:-1] + 1])
        before = torch.where(starts > 0, c[(starts - 1).clamp(min=0)], torch.zeros_like(starts))
        h = (c[rec_end] - before) * torch.where(starts > 0, pinv[(starts - 1).clamp(min=0)], torch.ones_like(starts))
        count += int(rec_end.numel())
        s1 = (s1 + int(_mix(h, torch).sum())) & ((1 << 64) - 1)
        s2 = (s2 + int(_mix(h ^ _i64(0xD6E8FEB86659FD93), torch).sum())) & ((1 << 64) - 1)
        pos += used
        del seg, c, h, before, starts, nl, rec_end
    return count, s1, s2


def decode_shard(ctx, batch, read_len, phred, device, timings=None):
    """The finished single-end Batch's archive streams back to FASTQ text on the device (uint8 tensor).
    timings (a dict, optional) receives the wall seconds of the two device stages: 'ac_decode' (scalce_ac_decode: the coded
    stream, resident in HBM, back to q' symbols) and 'records' (scalce_fastq_records: records + names + symbols -> text;
    the read and name streams go in as host buffers, as a decompressor that has just read the files holds them)."""
    import ctypes as C
    import time

    import torch
    n = batch.n_reads
    nsym = n * read_len
    sym = torch.empty(nsym + 64, dtype=torch.uint8, device=device)
    p, nbytes = batch.output_ptr(host.OUT_QUAL, 0)
    table = batch.output(host.OUT_TABLE, 0, np.uint32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.ac_decode(table, p, nbytes, nsym, sym.data_ptr())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    reads = batch.output(host.OUT_READS, 0)
    names = batch.output(host.OUT_NAMES, 0)
    L = ctx.L
    cap = L.scalce_fastq_text_bytes(read_len, n, len(names), None)
    out = torch.empty(cap + 64, dtype=torch.uint8, device=device)
    nb = C.c_uint64(0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ctx._check(L.scalce_fastq_records(ctx.h, read_len, 1, reads.ctypes.data, len(reads), n, sym.data_ptr(), int(phred),
                                      names.ctypes.data, len(names), b"", 0, out.data_ptr(), cap, C.byref(nb), None, 0))
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(ac_decode=t1 - t0, records=t3 - t2, coded_bytes=int(nbytes), symbols=int(nsym), text_bytes=int(nb.value))
    return out[: nb.value]


def archive_hashes(batch, read_len, phred, n_reads):
    """SHA-256 of the three files a single-end Batch's streams make with the headers of compress.cpp:263-343 in front
    (scalce_amd/format.py): what `sha256sum PREFIX_1.scalce{n,r,q}` would print for this shard."""
    import hashlib
    import struct

    from . import format as fmt
    p = batch.params
    out = {}
    h = hashlib.sha256(fmt.MAGIC + struct.pack("<ii", p.no_ac, read_len))
    h.update(batch.output(host.OUT_READS, 0).tobytes())
    out["1.scalcer"] = h.hexdigest()
    h = hashlib.sha256(fmt.MAGIC + struct.pack("<q", phred))
    if not p.no_ac:
        h.update(batch.output(host.OUT_TABLE, 0).tobytes())
        h.update(struct.pack("<Q", n_reads * read_len))
    h.update(batch.output(host.OUT_QUAL, 0).tobytes())
    out["1.scalceq"] = h.hexdigest()
    h = hashlib.sha256(fmt.MAGIC + struct.pack("<B", 1 if p.use_names else 0))
    h.update(batch.output(host.OUT_NAMES, 0).tobytes())
    out["1.scalcen"] = h.hexdigest()
    return out


def sharded_records_text(comm, ctx, batch, res, read_len, phred, device):
    """One rank's part of a sharded run's self-check (bench.py at N > 1): the inverse of what scalce_sharded_compress did with
    the quality stream.  The rank decodes the coder blocks IT holds -- its range [sym_lo, sym_hi) of the run-wide reordered
    stream, coded against the run-wide table -- on the device; the symbols go back to the ranks whose records they belong to
    (the block-range plan run backwards: pieces out of the range, an all-to-all with send and receive sizes swapped), and
    every rank rebuilds the FASTQ text of the records it emitted from its own read and name streams and the symbols that came
    back.  Returns that text (uint8 device tensor); the caller sums record_digest() of it over the ranks and compares with
    the sum over the inputs -- records change owner between ranks, the multiset of the run does not."""
    import ctypes as C
    import os
    import sys
    import time

    import torch
    t_mark = [time.perf_counter()]

    def mark(what):   # SCALCE_TRACE=1: where the check's time goes
        if os.environ.get("SCALCE_TRACE"):
            torch.cuda.synchronize()
            now = time.perf_counter()
            print("  [verify, rank %d] %-22s %8.2f s" % (comm.rank, what, now - t_mark[0]), file=sys.stderr, flush=True)
            t_mark[0] = now
    W, rank, nb1 = comm.world, comm.rank, int(res.nb1)
    counts = np.ctypeslib.as_array(res.counts, shape=(W * nb1,)).copy().reshape(W, nb1)
    send, recv, lo, hi, psrc, pdst = host.shard_plan_blocks(W, rank, nb1, counts, read_len)
    assert (lo, hi) == (int(res.sym_lo[0]), int(res.sym_hi[0])), "block-range plan differs from the run's"
    n_local = int(res.reads_local)
    assert int(counts[rank].sum()) == n_local
    nmine = hi - lo
    sym = torch.empty(nmine + 64, dtype=torch.uint8, device=device)
    if nmine:
        p, nbytes = batch.output_ptr(host.OUT_QUAL, 0)
        ctx.ac_decode(batch.output(host.OUT_TABLE, 0, np.uint32), p, nbytes, nmine, sym.data_ptr())
    mark("decode my blocks")
    # pieces of my range -> what every source rank sent, source-major (the layout the forward all-to-all delivered)
    recv_total = int(recv.sum())
    assert recv_total == nmine
    got = torch.empty(recv_total + 64, dtype=torch.uint8, device=device)
    if len(psrc):
        order = np.argsort(pdst, kind="stable")     # copy_pieces wants its pieces in source order: here the range itself
        src_off = torch.from_numpy(pdst[order].astype(np.int64)).to(device)
        dst_off = torch.from_numpy(psrc[order].astype(np.int64)).to(device)
        ctx.copy_pieces(sym.data_ptr(), got.data_ptr(), src_off.data_ptr(), dst_off.data_ptr(), len(order), recv_total)
    torch.cuda.synchronize()
    del sym
    mark("pieces out of my range")
    back = torch.empty(n_local * read_len + 64, dtype=torch.uint8, device=device)
    assert int(send.sum()) == n_local * read_len
    comm.all_to_all_v(got.data_ptr(), recv, back.data_ptr(), send)   # sizes swapped: the way back
    torch.cuda.synchronize()
    del got
    mark("symbols back to owners")
    reads = batch.output(host.OUT_READS, 0)
    names = batch.output(host.OUT_NAMES, 0)
    L = ctx.L
    cap = L.scalce_fastq_text_bytes(read_len, n_local, len(names), None)
    out = torch.empty(cap + 64, dtype=torch.uint8, device=device)
    nb = C.c_uint64(0)
    if n_local:
        ctx._check(L.scalce_fastq_records(ctx.h, read_len, 1, reads.ctypes.data, len(reads), n_local, back.data_ptr(), int(phred),
                                          names.ctypes.data, len(names), b"", 0, out.data_ptr(), cap, C.byref(nb), None, 0))
        torch.cuda.synchronize()
    mark("records -> text")
    return out[: nb.value]


def digest_sum(digests):
    """Sum of (records, s1, s2) triples modulo 2^64: the digest of the union of the texts."""
    m = (1 << 64) - 1
    return (sum(d[0] for d in digests), sum(d[1] for d in digests) & m, sum(d[2] for d in digests) & m)
