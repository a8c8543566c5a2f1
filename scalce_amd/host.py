"""ctypes host over include/scalce_hip.h.

Names follow the reference's seam: a Context carries the core table
(read_patterns / prepare_aho_automata), a Batch is one FASTQ shard resident in HBM, and its stage
methods are the batched forms of aho_search / output_read / output_quality / aho_trie_bucket /
aho_output / set_ac_stat / ac_write.  PyTorch is only used by callers for device memory; this
module takes raw device pointers.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OUT_READS, OUT_NAMES, OUT_QUAL, OUT_TABLE, OUT_FREQ4, OUT_TOKENS, OUT_PERM, OUT_QSTREAM, OUT_BUCKET_COUNTS, OUT_QINPUT, OUT_NAMELEN = range(11)
STAGES = ("ingest", "quality", "tokenize", "order", "emit", "entropy")
ROOT_CORE = 0x3FFFFFFF


class ScalceError(RuntimeError):
    pass


def library_path():
    return os.path.join(_HERE, "lib", "libscalce_hip.so")


class QMap(C.Structure):
    _fields_ = [("offset", C.c_int32), ("values", C.c_int32 * 128)]


class Params(C.Structure):
    _fields_ = [("read_len", C.c_int32 * 2), ("paired", C.c_int32), ("use_names", C.c_int32), ("no_ac", C.c_int32),
                ("qmap", QMap * 2), ("bucket_set_size", C.c_uint64), ("qprev", (C.c_uint32 * 2) * 2)]


class ShardResult(C.Structure):
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("nb1", C.c_uint32), ("rounds", C.c_uint32), ("sweeps", C.c_uint32),
                ("chunks_total", C.c_uint32), ("reads_total", C.c_uint64), ("first_read", C.c_uint64), ("reads_local", C.c_uint64),
                ("moved_in", C.c_uint64 * 2), ("counts", C.POINTER(C.c_uint64)), ("name_bytes", C.POINTER(C.c_uint64)),
                ("sym_lo", C.c_uint64 * 2), ("sym_hi", C.c_uint64 * 2), ("coded_bytes", (C.c_uint64 * 64) * 2),
                ("keep", C.c_void_p * 8), ("keep_bytes", C.c_uint64 * 8), ("magic", C.c_uint32)]


SHARD_PREPARE_ONLY, SHARD_CODER_ASYNC = 1, 2


def lib():
    """Load the HIP library; fail loudly when it has not been built (no fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ScalceError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). scalce_amd has no CPU implementation.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64; if the
    # system copy gets loaded first (through this library's NEEDED entry) torch later finds no GPU.  Loading
    # torch first makes the dynamic linker resolve our libamdhip64.so.7 to the copy torch already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
    L.scalce_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.scalce_ctx_destroy.argtypes = [vp]
    L.scalce_last_error.argtypes = [vp]
    L.scalce_last_error.restype = C.c_char_p
    L.scalce_patterns_load_bin.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.scalce_patterns_load_text.argtypes = [vp, C.c_char_p, C.c_size_t]
    for f in ("scalce_patterns_count", "scalce_patterns_states", "scalce_patterns_buckets"):
        getattr(L, f).argtypes = [vp]
    L.scalce_pattern_length.argtypes = [vp, i32]
    L.scalce_pattern_string.argtypes = [vp, i32]
    L.scalce_pattern_string.restype = C.c_char_p
    L.scalce_qmap_init.argtypes = [C.POINTER(QMap), C.POINTER(C.c_int32), i32]
    L.scalce_qmap_init.restype = None
    L.scalce_params_default.argtypes = [C.POINTER(Params)]
    L.scalce_params_default.restype = None
    L.scalce_batch_create.argtypes = [vp, C.POINTER(Params), u64, u64, C.POINTER(vp)]
    L.scalce_batch_destroy.argtypes = [vp]
    L.scalce_workspace_create.argtypes = [vp, C.POINTER(vp)]
    L.scalce_workspace_destroy.argtypes = [vp]
    L.scalce_workspace_destroy.restype = None
    L.scalce_batch_create_shared.argtypes = [vp, C.POINTER(Params), u64, u64, vp, C.POINTER(vp)]
    L.scalce_batch_ingest.argtypes = [vp, i32, vp, u64, vp]
    L.scalce_batch_append.argtypes = [vp, vp, u64, vp, u64, i32, C.POINTER(u64), vp]
    L.scalce_batch_reset.argtypes = [vp]
    L.scalce_batch_set_lean.argtypes = [vp, i32]
    L.scalce_batch_set_lean.restype = None
    L.scalce_batch_quality.argtypes = [vp, vp]
    L.scalce_batch_tokenize.argtypes = [vp, vp, vp]
    L.scalce_batch_order.argtypes = [vp, vp]
    L.scalce_batch_emit.argtypes = [vp, vp]
    L.scalce_batch_entropy.argtypes = [vp, vp, vp]
    L.scalce_batch_entropy_begin.argtypes = [vp, vp, vp]
    L.scalce_batch_entropy_end.argtypes = [vp, vp]
    L.scalce_batch_entropy_stream_begin.argtypes = [vp, i32, vp, vp, u64, vp]
    L.scalce_batch_entropy_stream_prepare.argtypes = [vp, i32, vp, vp, u64, vp]
    L.scalce_batch_entropy_begin_group.argtypes = [C.POINTER(vp), i32, vp, vp]
    L.scalce_batch_entropy_begin_group_last.argtypes = [C.POINTER(vp), i32, vp, vp, i32]
    L.scalce_batch_compress.argtypes = [vp, vp, u64, vp, u64, vp]
    L.scalce_batch_front.argtypes = [vp, vp, u64, vp, u64, vp]
    L.scalce_batch_finish.argtypes = [vp, vp]
    L.scalce_batch_output.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(u64)]
    L.scalce_batch_set_frame_on_demand.argtypes = [vp, i32]
    L.scalce_batch_qual_bytes.argtypes = [vp, i32, C.POINTER(u64)]
    L.scalce_batch_qual_window.argtypes = [vp, i32, u64, u64, vp, vp]
    L.scalce_batch_reads.argtypes = [vp]
    L.scalce_batch_reads.restype = u64
    L.scalce_batch_stage_ms.argtypes = [vp, i32, C.POINTER(C.c_float), C.POINTER(i32)]
    L.scalce_batch_stage_reset.argtypes = [vp, i32]
    L.scalce_batch_stage_reset.restype = None
    L.scalce_batch_stats.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.scalce_batch_kernel_timing.argtypes = [vp, i32]
    L.scalce_batch_kernel_timing.restype = None
    L.scalce_batch_kernel_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(u64), C.POINTER(u64)]
    L.scalce_memcpy_d2h.argtypes = [vp, vp, vp, u64]
    L.scalce_memcpy_h2d.argtypes = [vp, vp, vp, u64]
    L.scalce_memcpy_d2d.argtypes = [vp, vp, vp, u64, vp]
    L.scalce_batch_tokenize_begin.argtypes = [vp, vp]
    L.scalce_batch_tokenize_end.argtypes = [vp, vp]
    L.scalce_batch_tokenize_settle.argtypes = [vp, vp, vp]
    L.scalce_batch_set_chunks.argtypes = [vp, C.POINTER(u64), C.c_uint32]
    L.scalce_batch_entropy_stream.argtypes = [vp, i32, vp, vp, u64, vp]
    L.scalce_ac_scale.argtypes = [vp, vp, C.c_uint32, vp, vp]
    L.scalce_batch_chunk_plan.argtypes = [vp, u64, C.POINTER(u64), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(u64), vp]
    L.scalce_batch_text_offset.argtypes = [vp, i32, u64, C.POINTER(u64), vp]
    L.scalce_comm_unique_id.argtypes = [C.c_char_p]
    L.scalce_comm_create_rccl.argtypes = [i32, i32, i32, C.c_char_p, C.POINTER(vp)]
    L.scalce_comm_create_shm.argtypes = [i32, i32, i32, C.c_char_p, u64, C.POINTER(vp)]
    L.scalce_comm_destroy.argtypes = [vp]
    L.scalce_comm_error.argtypes = [vp]
    L.scalce_comm_error.restype = C.c_char_p
    L.scalce_comm_barrier.argtypes = [vp, vp]
    L.scalce_comm_all_gather.argtypes = [vp, vp, vp, u64, vp]
    L.scalce_comm_all_reduce_sum_u64.argtypes = [vp, vp, u64, vp]
    L.scalce_comm_all_to_all_v.argtypes = [vp, vp, C.POINTER(u64), vp, C.POINTER(u64), vp]
    L.scalce_sharded_compress.argtypes = [vp, vp, vp, vp, u64, vp, u64, i32, vp, vp, C.POINTER(ShardResult)]
    L.scalce_shard_result_free.argtypes = [C.POINTER(ShardResult)]
    L.scalce_shard_result_free.restype = None
    L.scalce_copy_pieces.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, u64, vp]
    L.scalce_patterns_describe_host.argtypes = [C.c_char_p, C.c_size_t, i32, vp, C.c_size_t, C.POINTER(C.c_int32),
                                                C.POINTER(C.c_int32)]
    L.scalce_selftest_ac.argtypes = [vp, u64, C.c_uint32, i32, C.POINTER(C.c_uint32)]
    L.scalce_ac_decode.argtypes = [vp, vp, vp, u64, u64, vp, vp]
    L.scalce_fastq_text_bytes.restype = u64
    L.scalce_fastq_text_bytes.argtypes = [i32, u64, u64, C.c_char_p]
    L.scalce_fastq_records.argtypes = [vp, i32, i32, vp, u64, u64, vp, C.c_int64, vp, u64, C.c_char_p, i32, vp, u64,
                                       C.POINTER(u64), vp, vp]
    _LIB = L
    return L


def qmap_init(stat, lossy_percentage):
    """quality_mapping_init after sampling (qualities.cpp:99-174) -> (offset, values[128])."""
    q = QMap()
    st = (C.c_int32 * 128)(*[int(x) for x in stat[:128]])
    lib().scalce_qmap_init(C.byref(q), st, int(lossy_percentage))
    return q.offset, np.array(list(q.values), dtype=np.int32)


class Context:
    def __init__(self, device=0, patterns_bin=None, patterns_text=None):
        self.L = lib()
        self.h = C.c_void_p()
        self.device = int(device)
        self._table = (patterns_bin, 0) if patterns_bin is not None else (patterns_text, 1)
        rc = self.L.scalce_ctx_create(int(device), C.byref(self.h))
        if rc:
            msg = self.L.scalce_last_error(self.h).decode() if self.h else "scalce_ctx_create failed"
            raise ScalceError(msg)
        if patterns_bin is not None:
            self._check(self.L.scalce_patterns_load_bin(self.h, patterns_bin, len(patterns_bin)))
        elif patterns_text is not None:
            self._check(self.L.scalce_patterns_load_text(self.h, patterns_text, len(patterns_text)))

    def _check(self, rc):
        if rc:
            raise ScalceError(f"[{rc}] " + self.L.scalce_last_error(self.h).decode())

    @property
    def n_patterns(self):
        return self.L.scalce_patterns_count(self.h)

    @property
    def n_states(self):
        return self.L.scalce_patterns_states(self.h)

    @property
    def n_buckets(self):
        return self.L.scalce_patterns_buckets(self.h)

    def pattern(self, p):
        return self.L.scalce_pattern_string(self.h, p)

    def to_host(self, d_ptr, nbytes, dtype=np.uint8):
        out = np.empty(int(nbytes), dtype=np.uint8)
        if nbytes:
            self._check(self.L.scalce_memcpy_d2h(self.h, out.ctypes.data, d_ptr, int(nbytes)))
        return out.view(dtype)

    def selftest_ac(self, ncases=1 << 24, seed=1, general=False):
        out = (C.c_uint32 * 6)()
        self._check(self.L.scalce_selftest_ac(self.h, int(ncases), int(seed), int(general), out))
        return list(out)

    def bucket_patterns(self):
        """File-order core index of every bucket in emission order, root (0x3FFFFFFF) last."""
        blob, is_text = self._table
        out = np.zeros(self.n_buckets + 1, dtype=np.int32)
        ns, nb = C.c_int32(), C.c_int32()
        self._check(self.L.scalce_patterns_describe_host(blob, len(blob), is_text, out.ctypes.data, len(out),
                                                         C.byref(ns), C.byref(nb)))
        return out

    def copy_pieces(self, d_src, d_dst, d_piece_src, d_piece_dst, npieces, total, stream=0):
        self._check(self.L.scalce_copy_pieces(self.h, d_src, d_dst, d_piece_src, d_piece_dst, int(npieces), int(total), stream))

    def ac_scale(self, d_counters, factor, d_table, stream=0):
        """compress.cpp:297-313 on the device: table = max(1, (1 + counters) / factor)."""
        self._check(self.L.scalce_ac_scale(self.h, d_counters, int(factor), d_table, stream))

    def copy_d2d(self, dst, src, nbytes, stream=0):
        self._check(self.L.scalce_memcpy_d2d(self.h, dst, src, int(nbytes), stream))

    def fastq_records(self, read_len, reads_payload, nrecords, d_qual, phred, names_payload=None, library=None, has_buckets=True,
                      mate_digit=0, stream=0):
        """Records back to FASTQ text on the device (scalce_fastq_records); returns the text as bytes."""
        import torch
        reads = np.frombuffer(reads_payload, dtype=np.uint8)
        names = None if names_payload is None else np.frombuffer(names_payload, dtype=np.uint8)
        lib = None if library is None else library.encode()
        cap = self.L.scalce_fastq_text_bytes(read_len, nrecords, 0 if names is None else len(names), None if names is not None else lib)
        out = torch.empty(cap + 64, dtype=torch.uint8, device=f"cuda:{self.device}")
        nb = C.c_uint64(0)
        self._check(self.L.scalce_fastq_records(self.h, read_len, int(has_buckets), reads.ctypes.data, len(reads), nrecords, d_qual,
                                                int(phred), None if names is None else names.ctypes.data,
                                                0 if names is None else len(names), lib or b"", mate_digit, out.data_ptr(), cap,
                                                C.byref(nb), None, stream))
        return out[: nb.value].cpu().numpy().tobytes()

    def ac_decode(self, table_u32, d_blocks, nbytes, nsym, d_out, stream=0):
        t = np.ascontiguousarray(table_u32, dtype=np.uint32)
        self._check(self.L.scalce_ac_decode(self.h, t.ctypes.data, d_blocks, int(nbytes), int(nsym), d_out, stream))

    def close(self):
        if self.h:
            self.L.scalce_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """Collectives of a sharded run (scalce_comm): RCCL between processes with one GPU each, or the shared-memory
    rehearsal transport between processes that share a GPU."""

    def __init__(self, device, world, rank, unique_id=None, shm_name=None, slot_bytes=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.world, self.rank = world, rank
        if shm_name is not None:
            rc = self.L.scalce_comm_create_shm(device, world, rank, shm_name.encode(), int(slot_bytes), C.byref(self.h))
        else:
            rc = self.L.scalce_comm_create_rccl(device, world, rank, unique_id, C.byref(self.h))
        if rc:
            raise ScalceError(f"[{rc}] " + (self.L.scalce_comm_error(self.h) or b"").decode())

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        if lib().scalce_comm_unique_id(buf):
            raise ScalceError("RCCL is not available (ncclGetUniqueId)")
        return buf.raw

    def _check(self, rc):
        if rc:
            raise ScalceError(f"[{rc}] " + self.L.scalce_comm_error(self.h).decode())

    def barrier(self, stream=0):
        self._check(self.L.scalce_comm_barrier(self.h, stream))

    def all_gather(self, d_send, d_recv, nbytes, stream=0):
        self._check(self.L.scalce_comm_all_gather(self.h, d_send, d_recv, int(nbytes), stream))

    def all_reduce_sum_u64(self, d_buf, count, stream=0):
        self._check(self.L.scalce_comm_all_reduce_sum_u64(self.h, d_buf, int(count), stream))

    def all_to_all_v(self, d_send, send_bytes, d_recv, recv_bytes, stream=0):
        """send_bytes[d] bytes to every rank d (consecutive ranges of d_send), recv_bytes[src] from every rank src."""
        sb = (C.c_uint64 * self.world)(*[int(x) for x in send_bytes])
        rb = (C.c_uint64 * self.world)(*[int(x) for x in recv_bytes])
        self._check(self.L.scalce_comm_all_to_all_v(self.h, d_send, sb, d_recv, rb, stream))

    def close(self):
        if self.h:
            self.L.scalce_comm_destroy(self.h)
            self.h = C.c_void_p()


def sharded_compress(comm, ctx, batch, d_text1, n1, d_text2=None, n2=0, flags=0, stream=0, coder_stream=0, result=None):
    """SPMD body of a sharded run (scalce_sharded_compress); returns the ShardResult (free it with shard_result_free).
    `result`: the result of an earlier call on the same batch -- its device buffers are reused."""
    res = result if result is not None else ShardResult()
    rc = ctx.L.scalce_sharded_compress(comm.h, ctx.h, batch.h, d_text1, int(n1), d_text2, int(n2), int(flags), stream, coder_stream,
                                       C.byref(res))
    if rc:
        raise ScalceError(f"[{rc}] sharded run failed on rank {comm.rank} (message on stderr)")
    return res


def shard_plan_blocks(world, rank, nb1, counts, read_len):
    """scalce_shard_plan_blocks: the run-wide reordered quality stream dealt out in ranges of whole 10 MiB blocks.
    counts = [world][nb1] reads per (rank, bucket).  Returns (send_bytes, recv_bytes, lo, hi, piece_src, piece_dst)."""
    L = lib()
    u64p = C.POINTER(C.c_uint64)
    L.scalce_shard_plan_blocks.argtypes = [C.c_int, C.c_int, C.c_uint32, u64p, C.c_uint64, u64p, u64p, u64p, u64p, u64p, u64p, u64p]
    flat = np.ascontiguousarray(counts, dtype=np.uint64).reshape(-1)
    send, recv = np.zeros(world, np.uint64), np.zeros(world, np.uint64)
    ps, pd = np.zeros(world * nb1 + 1, np.uint64), np.zeros(world * nb1 + 1, np.uint64)
    lo, hi, npc = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = L.scalce_shard_plan_blocks(world, rank, nb1, flat.ctypes.data_as(u64p), int(read_len), send.ctypes.data_as(u64p),
                                    recv.ctypes.data_as(u64p), C.byref(lo), C.byref(hi), ps.ctypes.data_as(u64p), pd.ctypes.data_as(u64p),
                                    C.byref(npc))
    if rc:
        raise ScalceError(f"[{rc}] scalce_shard_plan_blocks")
    return send, recv, int(lo.value), int(hi.value), ps[: npc.value].copy(), pd[: npc.value].copy()


def shard_result_free(res):
    lib().scalce_shard_result_free(C.byref(res))


def entropy_begin_group(batches, prep_stream=0, stream=0, last=False):
    """ONE coder launch over the blocks of several shards (scalce_batch_entropy_begin_group); each shard is completed
    by its own entropy_end / finish on `stream`.  last: 1 / True = nothing will be queued behind this launch (the end of a
    run); 2 = a small launch at the start of a run (the kernel that holds the fewest CUs whatever the size)."""
    arr = (C.c_void_p * len(batches))(*[b.h for b in batches])
    batches[0]._check(batches[0].L.scalce_batch_entropy_begin_group_last(arr, len(batches), prep_stream, stream, int(last)))


class StreamStats(C.Structure):
    _fields_ = [("total_s", C.c_double), ("read_wait_s", C.c_double), ("h2d_wait_s", C.c_double), ("front_s", C.c_double),
                ("order_s", C.c_double), ("emit_s", C.c_double), ("entropy_s", C.c_double), ("rounds", C.c_uint64),
                ("reads", C.c_uint64), ("bytes", C.c_uint64 * 2)]


READ_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_void_p, C.c_uint64)
STREAM_LEAN, STREAM_DEFER_ENTROPY = 1, 2


def stream_compress(ctx, params, read1, read2=None, piece_bytes=0, reads_hint=0, flags=0):
    """scalce_stream_compress: read1 / read2 are callables (cap) -> bytes (b"" at the end of the stream; raise for a read
    error), called from the library's reader threads.  Returns (Batch holding the results, StreamStats)."""
    def wrap(fn):
        def cb(_user, dst, cap):
            try:
                data = fn(int(cap))
            except Exception:  # noqa: BLE001 - reported to the library as a read error
                return -1
            if data:
                C.memmove(dst, data, len(data))
            return len(data)
        return READ_FN(cb)
    cb1 = wrap(read1)
    cb2 = wrap(read2) if read2 is not None else C.cast(None, READ_FN)
    out, st = C.c_void_p(), StreamStats()
    msg = C.create_string_buffer(512)
    L = ctx.L
    L.scalce_stream_compress.argtypes = [C.c_void_p, C.POINTER(Params), READ_FN, C.c_void_p, READ_FN, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int,
                                         C.POINTER(C.c_void_p), C.POINTER(StreamStats), C.c_char_p, C.c_size_t]
    rc = L.scalce_stream_compress(ctx.h, C.byref(params), cb1, None, cb2, None, int(piece_bytes), int(reads_hint), int(flags), C.byref(out),
                                  C.byref(st), msg, len(msg))
    if rc:
        raise ScalceError(f"[{rc}] " + msg.value.decode(errors="replace"))
    b = Batch.__new__(Batch)
    b.ctx, b.L, b.params, b.h, b.workspace = ctx, L, params, out, None
    return b, st


class Workspace:
    """Front-stage device buffers shared by several batches (scalce_workspace): see include/scalce_hip.h."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx._check(ctx.L.scalce_workspace_create(ctx.h, C.byref(self.h)))

    def close(self):
        if self.h:
            self.ctx.L.scalce_workspace_destroy(self.h)
            self.h = C.c_void_p()


class Batch:
    """One FASTQ shard in HBM (scalce_batch)."""

    def __init__(self, ctx, read_len, max_reads, max_text, paired=False, use_names=True, no_ac=False, qmap=None,
                 bucket_set_size=0, read_len2=0, qprev=None, workspace=None):
        self.ctx = ctx
        self.L = ctx.L
        p = Params()
        self.L.scalce_params_default(C.byref(p))
        p.read_len[0] = int(read_len)
        p.read_len[1] = int(read_len2 or read_len) if paired else 0
        p.paired, p.use_names, p.no_ac = int(paired), int(use_names), int(no_ac)
        p.bucket_set_size = int(bucket_set_size)
        if qmap is not None:  # [(offset, values)] per mate
            for m, (off, vals) in enumerate(qmap):
                p.qmap[m].offset = int(off)
                for i in range(128):
                    p.qmap[m].values[i] = int(vals[i])
        if qprev is not None:
            for m in range(2):
                for i in range(2):
                    p.qprev[m][i] = int(qprev[m][i])
        self.params = p
        self.h = C.c_void_p()
        self.workspace = workspace  # keeps it alive
        if workspace is not None:
            rc = self.L.scalce_batch_create_shared(ctx.h, C.byref(p), int(max_reads), int(max_text), workspace.h, C.byref(self.h))
        else:
            rc = self.L.scalce_batch_create(ctx.h, C.byref(p), int(max_reads), int(max_text), C.byref(self.h))
        if rc:
            raise ScalceError(f"[{rc}] " + self.L.scalce_last_error(ctx.h).decode())

    def _check(self, rc):
        self.ctx._check(rc)

    def ingest(self, mate, d_text, nbytes, stream=0):
        self._check(self.L.scalce_batch_ingest(self.h, mate, d_text, int(nbytes), stream))

    def append(self, d_text1, n1, d_text2=None, n2=0, final=False, stream=0):
        """Next piece of the read stream behind the rows already held; returns the bytes consumed per mate."""
        used = (C.c_uint64 * 2)()
        self._check(self.L.scalce_batch_append(self.h, d_text1, int(n1), d_text2, int(n2), int(final), used, stream))
        return used[0], used[1]

    def reset(self):
        self._check(self.L.scalce_batch_reset(self.h))

    def set_lean(self, lean=True):
        self.L.scalce_batch_set_lean(self.h, int(lean))

    def quality(self, stream=0):
        self._check(self.L.scalce_batch_quality(self.h, stream))

    def tokenize(self, d_prior_counts=None, stream=0):
        self._check(self.L.scalce_batch_tokenize(self.h, d_prior_counts, stream))

    def tokenize_begin(self, stream=0):
        self._check(self.L.scalce_batch_tokenize_begin(self.h, stream))

    def tokenize_end(self, stream=0):
        self._check(self.L.scalce_batch_tokenize_end(self.h, stream))

    def set_chunks(self, starts):
        a = (C.c_uint64 * len(starts))(*[int(x) for x in starts])
        self._check(self.L.scalce_batch_set_chunks(self.h, a, len(starts)))

    def entropy_stream(self, mate, d_table, d_symbols, nsym, stream=0):
        self._check(self.L.scalce_batch_entropy_stream(self.h, mate, d_table, d_symbols, int(nsym), stream))

    def order(self, stream=0):
        self._check(self.L.scalce_batch_order(self.h, stream))

    def emit(self, stream=0):
        self._check(self.L.scalce_batch_emit(self.h, stream))

    def entropy(self, d_table_override=None, stream=0):
        self._check(self.L.scalce_batch_entropy(self.h, d_table_override, stream))

    def entropy_begin(self, d_table_override=None, stream=0):
        """Enqueue the entropy stage and return; entropy_end / finish waits for it (several shards in flight)."""
        self._check(self.L.scalce_batch_entropy_begin(self.h, d_table_override, stream))

    def entropy_end(self, stream=0):
        self._check(self.L.scalce_batch_entropy_end(self.h, stream))

    def entropy_stream_begin(self, mate, d_table, d_symbols, nsym, stream=0):
        self._check(self.L.scalce_batch_entropy_stream_begin(self.h, mate, d_table, d_symbols, int(nsym), stream))

    def entropy_stream_prepare(self, mate, d_table, d_symbols, nsym, stream=0):
        self._check(self.L.scalce_batch_entropy_stream_prepare(self.h, mate, d_table, d_symbols, int(nsym), stream))

    def front(self, d_text1, n1, d_text2=None, n2=0, stream=0):
        """Every stage before the entropy coder (ingest .. emit) on `stream` (scalce_batch_front)."""
        self._check(self.L.scalce_batch_front(self.h, d_text1, int(n1), d_text2, int(n2), stream))

    def compress(self, d_text1, n1, d_text2=None, n2=0, stream=0):
        self._check(self.L.scalce_batch_compress(self.h, d_text1, int(n1), d_text2, int(n2), stream))

    def finish(self, stream=0):
        self._check(self.L.scalce_batch_finish(self.h, stream))

    @property
    def n_reads(self):
        return int(self.L.scalce_batch_reads(self.h))

    def output_ptr(self, which, mate=0):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self.L.scalce_batch_output(self.h, which, mate, C.byref(p), C.byref(n)))
        return p.value or 0, n.value

    def output(self, which, mate=0, dtype=np.uint8):
        p, n = self.output_ptr(which, mate)
        return self.ctx.to_host(p, n, dtype)

    def set_frame_on_demand(self, on=True):
        """The coded blocks are framed on their way out (qual_window) instead of by a copy pass behind the coder."""
        self._check(self.L.scalce_batch_set_frame_on_demand(self.h, int(on)))

    def set_code_in_place(self, on=True):
        """Grouped coder launches write a block's bytes over its own symbols (no block buffers); the caller keeps the shard's text
        in place until the shard is collected (scalce_batch_set_code_in_place)."""
        self._check(self.L.scalce_batch_set_code_in_place(self.h, int(on)))

    @property
    def reruns(self):
        self.L.scalce_batch_reruns.restype = C.c_uint64
        self.L.scalce_batch_reruns.argtypes = [C.c_void_p]
        return int(self.L.scalce_batch_reruns(self.h))

    def qual_bytes(self, mate=0):
        n = C.c_uint64()
        self._check(self.L.scalce_batch_qual_bytes(self.h, mate, C.byref(n)))
        return n.value

    def qual_window(self, mate, offset, nbytes, dst, stream=0):
        """bytes [offset, offset + nbytes) of SCALCE_OUT_QUAL into dst (device or pinned host pointer, 4-byte aligned)"""
        self._check(self.L.scalce_batch_qual_window(self.h, mate, int(offset), int(nbytes), dst, stream))

    def stats(self):
        a = (C.c_uint32 * 6)()
        self._check(self.L.scalce_batch_stats(self.h, a))
        return dict(tie_reads=a[0], events=a[1], jacobi_iters=a[2], chunks=a[3], order_run_members=a[4], tie_fallback=a[5])

    def stage_reset(self, enable=True):
        self.L.scalce_batch_stage_reset(self.h, int(enable))

    def stage_ms(self):
        out = {}
        for i, name in enumerate(STAGES):
            ms, n = C.c_float(), C.c_int()
            self.L.scalce_batch_stage_ms(self.h, i, C.byref(ms), C.byref(n))
            out[name] = (ms.value, n.value)
        return out

    def kernel_timing(self, enable=True):
        self.L.scalce_batch_kernel_timing(self.h, int(enable))

    def kernel_ms(self):
        ms, n, bi, bo = C.c_double(), C.c_int(), C.c_uint64(), C.c_uint64()
        self._check(self.L.scalce_batch_kernel_ms(self.h, C.byref(ms), C.byref(n), C.byref(bi), C.byref(bo)))
        return dict(total_ms=ms.value, launches=n.value, bytes_in=bi.value, bytes_out=bo.value)

    def close(self):
        if self.h:
            self.L.scalce_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
