"""Sharded runs: one FASTQ read range per GPU, one archive.

Semantics (DESIGN.md "Multi-GPU"): every shard behaves like one spill chunk of the reference
(compress.cpp:702-715 dumps, :104-159 merges): the archive lists buckets in emission order and, inside a
bucket, shard 0's records, then shard 1's, ... each shard sorted on its own.  Everything that the reference
carries across reads is carried across shards exactly:

* tie-break counts (`bin_size`, reads.cpp:246,420): a Jacobi fixed point over ALL ranks -- per sweep an
  all-gather of the per-bucket counts gives every rank the counts of the shards before it
  (scalce_batch_tokenize_sweep's prior);
* the order-2 quality model: all-reduce of the 80^3 counters plus the trigrams that straddle a shard boundary;
* the arithmetic coder's 10 MiB blocks are cut on the run-wide reordered stream: ranks own contiguous block
  ranges and receive the q' bytes of their range with one all-to-all (the local reordered stream of a rank is
  already sorted by run-wide offset, so the send side needs no packing).

`Comm` is the collective interface; `TorchComm` wraps torch.distributed (RCCL on GPUs, gloo for rehearsals),
`ThreadComm` runs virtual ranks as threads of one process (tests on a single GPU).
"""
import threading

import numpy as np

from . import host

AC_BLOCK = 10 * 1024 * 1024


# ----------------------------------------------------------------------------------------------- comms
class TorchComm:
    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.cpu_staged = dist.get_backend() == "gloo"

    def _stage(self, t):
        return t.cpu() if (self.cpu_staged and t.is_cuda) else t

    def all_gather(self, t):
        import torch
        x = self._stage(t.contiguous())
        out = [torch.empty_like(x) for _ in range(self.world)]
        self.dist.all_gather(out, x)
        return torch.stack(out).to(t.device)

    def all_reduce_sum(self, t):
        x = self._stage(t)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM)
        if x is not t:
            t.copy_(x)
        return t

    def all_reduce_max(self, value):
        import torch
        x = torch.tensor([int(value)], dtype=torch.int64)
        if not self.cpu_staged:
            x = x.cuda()
        self.dist.all_reduce(x, op=self.dist.ReduceOp.MAX)
        return int(x.item())

    def all_to_all(self, send, send_splits, recv_splits):
        import torch
        s = self._stage(send.contiguous())
        r = torch.empty(int(sum(recv_splits)), dtype=send.dtype, device=s.device)
        self.dist.all_to_all_single(r, s, [int(x) for x in recv_splits], [int(x) for x in send_splits])
        return r.to(send.device)

    def barrier(self):
        self.dist.barrier()


class _ThreadShared:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadComm:
    """Virtual ranks = threads of this process sharing one GPU (tests)."""

    def __init__(self, shared, rank):
        self.sh, self.rank, self.world = shared, rank, shared.world

    def _exchange(self, obj):
        self.sh.slots[self.rank] = obj
        self.sh.barrier.wait()
        out = list(self.sh.slots)
        self.sh.barrier.wait()
        return out

    def all_gather(self, t):
        import torch
        return torch.stack([x.clone() for x in self._exchange(t)])

    def all_reduce_sum(self, t):
        parts = self._exchange(t.clone())
        t.copy_(sum(parts[1:], parts[0].clone()))
        return t

    def all_reduce_max(self, value):
        return max(self._exchange(int(value)))

    def all_to_all(self, send, send_splits, recv_splits):
        import torch
        off = np.concatenate([[0], np.cumsum(send_splits)]).astype(np.int64)
        everyone = self._exchange((send, off))
        parts = [everyone[src][0][int(everyone[src][1][self.rank]):int(everyone[src][1][self.rank + 1])].clone()
                 for src in range(self.world)]
        assert [p.numel() for p in parts] == [int(x) for x in recv_splits]
        self.sh.barrier.wait()
        return torch.cat(parts) if parts else send[:0]

    def barrier(self):
        self.sh.barrier.wait()


def run_threads(world, fn):
    """Run fn(comm) on `world` virtual ranks; returns the list of results (rank order)."""
    shared = _ThreadShared(world)
    out, err = [None] * world, [None] * world

    def work(r):
        try:
            out[r] = fn(ThreadComm(shared, r))
        except BaseException as e:  # noqa: BLE001 - re-raised below
            err[r] = e
            shared.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out


# ----------------------------------------------------------------------------------------- plan (host math)
def block_ranges(total_symbols, world):
    """Contiguous 10 MiB-block ranges per rank -> symbol ranges [lo, hi)."""
    nblk = -(-int(total_symbols) // AC_BLOCK)
    lo = [min(int(total_symbols), (d * nblk // world) * AC_BLOCK) for d in range(world)]
    hi = [min(int(total_symbols), ((d + 1) * nblk // world) * AC_BLOCK) for d in range(world)]
    return lo, hi


def stream_plan(C, L, rank):
    """C[world, nb1] reads per (rank, bucket).  Returns what `rank` sends to / receives from everyone when the
    run-wide reordered quality stream (bucket-major, rank-major inside a bucket) is dealt out in block ranges."""
    C = np.asarray(C, dtype=np.int64)
    world, _ = C.shape
    Cg = C.sum(axis=0)
    bucket_base = (np.cumsum(Cg) - Cg) * L                       # run-wide offset of each bucket
    before = np.cumsum(C, axis=0) - C                             # reads of lower ranks in the same bucket
    g0 = bucket_base[None, :] + before * L                        # [world, nb1] run-wide start of every piece
    ln = C * L
    lo, hi = block_ranges(int(Cg.sum()) * L, world)

    def below(r, X):  # bytes of rank r's stream with run-wide offset < X
        return int(np.clip(X - g0[r], 0, ln[r]).sum())

    send = [below(rank, hi[d]) - below(rank, lo[d]) for d in range(world)]
    recv = [below(src, hi[rank]) - below(src, lo[rank]) for src in range(world)]
    # pieces of what this rank receives: contiguous in the receive buffer (source-major, bucket order)
    psrc, pdst = [], []
    base = 0
    for src in range(world):
        a = np.maximum(g0[src], lo[rank])
        b = np.minimum(g0[src] + ln[src], hi[rank])
        keep = b > a
        lens = (b - a)[keep]
        starts = base + np.cumsum(lens) - lens
        psrc.append(starts)
        pdst.append(a[keep] - lo[rank])
        base += int(lens.sum())
    assert base == sum(recv)
    return dict(send=send, recv=recv, piece_src=np.concatenate(psrc).astype(np.uint64),
                piece_dst=np.concatenate(pdst).astype(np.uint64), lo=lo[rank], hi=hi[rank], total=int(Cg.sum()) * L)


def boundary_trigrams(edges, nsym):
    """edges[r] = (first two, last two) q' symbols of shard r, nsym[r] its symbol count.  Trigram keys that
    straddle a shard boundary (the reference's prev[] runs across reads, qualities.cpp:179)."""
    seq = []  # the symbols adjacent to boundaries, with a shard id each
    for r, (e, n) in enumerate(zip(edges, nsym)):
        if n == 0:
            continue
        if n == 1:
            raise NotImplementedError("a shard with a single quality symbol")
        a, b, c, d = (int(x) for x in e)
        seq += [(r, 0, a), (r, 1, b)] if n == 2 else [(r, 0, a), (r, 1, b), (r, n - 2, c), (r, n - 1, d)]
    keys = []
    for i in range(2, len(seq)):
        (r0, p0, s0), (r1, p1, s1), (r2, p2, s2) = seq[i - 2], seq[i - 1], seq[i]
        if r0 == r2:
            continue  # inside one shard: already counted there (or not adjacent at all)
        adjacent = all(((ra == rb and pb == pa + 1) or (ra != rb and pb == 0 and pa == nsym[ra] - 1))
                       for (ra, pa, _), (rb, pb, _) in ((seq[i - 2], seq[i - 1]), (seq[i - 1], seq[i])))
        if adjacent and max(s0, s1, s2) < 80:
            keys.append((s0 * 80 + s1) * 80 + s2)
    return keys


# ------------------------------------------------------------------------------------------- the SPMD body
class ShardResult:
    pass


def compress_shard(comm, ctx, batch, d_text, nbytes, d_text2=None, nbytes2=0, stream=0, ent_stream=None, prepare_only=False):
    """SPMD: every rank calls this with its own shard.  Leaves the rank's pieces of the archive in `batch`
    (reads/names payload of the shard, AC blocks of the rank's block range) and returns the metadata needed to
    assemble or to report.

    `stream` (raw handle) must be torch's current stream.  With `ent_stream` (a torch.cuda.Stream) the arithmetic
    coder is only enqueued there, behind everything issued so far, and the call returns while it runs: the
    caller owes a `batch.finish(ent_stream.cuda_stream)` before it reads the result or reuses `batch`, and can
    start the next shard's front stages (and their collectives, still issued in program order on every rank)
    in the meantime.  With `prepare_only` the coder is not even enqueued: the shard's range of the run-wide stream and
    the run-wide table are handed to the batch (entropy_stream_prepare) and the caller codes several shards with one
    launch (host.entropy_begin_group), then finishes each."""
    import torch
    dev = torch.device("cuda", ctx_device(ctx))
    p = batch.params
    nm = 2 if p.paired else 1
    L = [p.read_len[0], p.read_len[1]]
    batch.ingest(0, d_text, nbytes, stream)
    if nm == 2:
        batch.ingest(1, d_text2, nbytes2, stream)
    batch.quality(stream)
    batch.finish(stream)
    n = batch.n_reads
    res = ShardResult()
    res.n_reads = n
    n_all = comm.all_gather(torch.tensor([n], dtype=torch.int64, device=dev)).cpu().numpy().reshape(-1)
    res.n_all = n_all
    # ---- run-wide quality model
    tables = []
    if not p.no_ac:
        for m in range(nm):
            f4 = torch.empty(512000, dtype=torch.int64, device=dev)
            ptr, nb = batch.output_ptr(host.OUT_FREQ4, m)
            ctx.copy_d2d(f4.data_ptr(), ptr, nb, stream)
            comm.all_reduce_sum(f4)
            qp, qn = batch.output_ptr(host.OUT_QINPUT, m)
            e = torch.zeros(4, dtype=torch.uint8, device=dev)
            if qn >= 2:
                ctx.copy_d2d(e.data_ptr(), qp, 2, stream)
                ctx.copy_d2d(e.data_ptr() + 2, qp + qn - 2, 2, stream)
            torch.cuda.current_stream().synchronize()
            edges = comm.all_gather(e).cpu().numpy()
            keys = boundary_trigrams(edges, [int(x) * L[m] for x in n_all])
            if keys:
                k = torch.tensor(keys, dtype=torch.int64, device=dev)
                f4.index_add_(0, k, torch.ones_like(k))
            factor = 1 + (int(n_all.sum()) * L[m]) // 0xFFFFFFFF  # compress.cpp:297-303 on the run-wide count
            tables.append(torch.clamp_min((f4 + 1) // factor, 1).to(torch.int32))  # same bits as u32
    res.tables = tables
    # ---- tokenize: Jacobi fixed point over all ranks
    batch.tokenize_begin(stream)
    sweeps = 0
    while True:
        cp, cn = batch.output_ptr(host.OUT_BUCKET_COUNTS, 0)
        counts = torch.empty(cn // 8, dtype=torch.int64, device=dev)
        ctx.copy_d2d(counts.data_ptr(), cp, cn, stream)
        torch.cuda.current_stream().synchronize()
        allc = comm.all_gather(counts)                       # [world, nb1]
        prior = allc[:comm.rank].sum(dim=0) if comm.rank else torch.zeros_like(counts)
        changed = batch.tokenize_sweep(prior.data_ptr(), stream)
        sweeps += 1
        if comm.all_reduce_max(changed) == 0:
            break
    batch.tokenize_end(stream)
    res.sweeps = sweeps
    batch.order(stream)
    batch.emit(stream)
    batch.finish(stream)
    cp, cn = batch.output_ptr(host.OUT_BUCKET_COUNTS, 0)
    counts = torch.empty(cn // 8, dtype=torch.int64, device=dev)
    ctx.copy_d2d(counts.data_ptr(), cp, cn, stream)
    torch.cuda.current_stream().synchronize()
    C = comm.all_gather(counts).cpu().numpy()
    res.C = C
    # ---- run-wide block ranges of the reordered quality stream, all-to-all, code
    res.plans = []
    keep = []
    if not p.no_ac:
        for m in range(nm):
            plan = stream_plan(C, L[m], comm.rank)
            qsp, qsn = batch.output_ptr(host.OUT_QSTREAM, m)
            local = torch.empty(qsn, dtype=torch.uint8, device=dev)
            ctx.copy_d2d(local.data_ptr(), qsp, qsn, stream)
            torch.cuda.current_stream().synchronize()
            got = comm.all_to_all(local, plan["send"], plan["recv"])
            mine = torch.empty(plan["hi"] - plan["lo"] + 16, dtype=torch.uint8, device=dev)
            if got.numel():
                ps = torch.from_numpy(plan["piece_src"].astype(np.int64)).to(dev)
                pd = torch.from_numpy(plan["piece_dst"].astype(np.int64)).to(dev)
                ctx.copy_pieces(got.data_ptr(), mine.data_ptr(), ps.data_ptr(), pd.data_ptr(), ps.numel(), got.numel(), stream)
            if prepare_only:
                batch.entropy_stream_prepare(m, tables[m].data_ptr(), mine.data_ptr(), plan["hi"] - plan["lo"], stream)
            elif ent_stream is None:
                batch.entropy_stream(m, tables[m].data_ptr(), mine.data_ptr(), plan["hi"] - plan["lo"], stream)
                batch.finish(stream)
            else:
                ent_stream.wait_stream(torch.cuda.current_stream())
                batch.entropy_stream_begin(m, tables[m].data_ptr(), mine.data_ptr(), plan["hi"] - plan["lo"],
                                           ent_stream.cuda_stream)
            # only what the coder reads after this call returns; `local` and `got` were consumed on this stream
            keep.append((mine, tables[m]))
            res.plans.append(plan)
    batch._keep_alive = keep  # read by the coder after this call returns
    return res


def ctx_device(ctx):
    return getattr(ctx, "device", 0)


# ----------------------------------------------------------------------------------------- assembly (host)
def assemble(ctx, results, batches, L, sz_meta=None):
    """Put the ranks' pieces together into the payloads of ONE archive (host side, numpy).  results/batches in
    rank order.  Returns dict(reads=bytes, names=bytes, qual=bytes, table=np.uint32[512000])."""
    world = len(results)
    C = results[0].C
    nb1 = C.shape[1]
    order = ctx.bucket_patterns()
    levels = np.array([0 if p == host.ROOT_CORE else len(ctx.pattern(int(p))) for p in order], dtype=np.int64)
    sz_meta = (2 if L > 255 else 1) if sz_meta is None else sz_meta
    recsz = (L - levels + 3) // 4 + sz_meta
    reads_r = [b.output(host.OUT_READS, 0) for b in batches]
    off_r = []
    for r in range(world):
        sz = np.where(C[r] > 0, 12 + C[r] * recsz, 0)
        off_r.append(np.cumsum(sz) - sz)
    out = []
    Cg = C.sum(axis=0)
    import struct
    for b in np.flatnonzero(Cg):
        out.append(struct.pack("<iq", int(order[b]), int(Cg[b])))
        for r in range(world):
            if C[r][b]:
                a = int(off_r[r][b]) + 12
                out.append(reads_r[r][a:a + int(C[r][b] * recsz[b])].tobytes())
    reads = b"".join(out)
    names = b""
    if batches[0].params.use_names:
        pieces = []
        per = []
        for r in range(world):
            nl = batches[r].output(host.OUT_NAMELEN, 0).astype(np.int64)
            perm = batches[r].output(host.OUT_PERM, 0, np.uint32)
            rec = 1 + nl[perm]
            ends = np.cumsum(rec)
            first = np.cumsum(C[r]) - C[r]
            start_byte = np.concatenate([[0], ends])[first]
            end_byte = np.concatenate([[0], ends])[first + C[r]]
            per.append((batches[r].output(host.OUT_NAMES, 0), start_byte, end_byte))
        for b in np.flatnonzero(Cg):
            for r in range(world):
                if C[r][b]:
                    buf, sb, eb = per[r]
                    pieces.append(buf[int(sb[b]):int(eb[b])].tobytes())
        names = b"".join(pieces)
    qual = b"".join(b.output(host.OUT_QUAL, 0).tobytes() for b in batches)
    table = batches[0].output(host.OUT_TABLE, 0, np.uint32) if not batches[0].params.no_ac else None
    return dict(reads=reads, names=names, qual=qual, table=table)
