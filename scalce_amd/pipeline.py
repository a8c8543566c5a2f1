"""Shards in flight: binding of the C ABI's scalce_pipeline_* (scalce_amd/csrc/pipeline.cpp).

The reference keeps its cores busy by handing `-T` blocks per batch to coder threads while the reader goes on
(arithmetic.cpp:349-357, compress.cpp:781-786).  The device-side counterpart: the arithmetic coder is a long kernel that only
depends on its own shard's front stages, so the front stages (ingest .. emit, and in a sharded run their collectives) of the
NEXT shards run on one stream beside the coder launches of the previous ones on `coder_streams` others, `group` shards per
launch, shards retired on events.  The loop itself -- slots, streams, events, which kernel the last launch of a run takes --
is C++ behind the C ABI (rounds 1-3: Python, here); this class only turns its flags into the callbacks bench.py and the
tests use.

Usage:
    pipe = ShardPipeline(batches, group=3)
    for shard in shards:
        slot, batch = pipe.acquire()          # waits for the batch's previous shard, calls on_retire for it
        ... front stages of `shard` into `batch` on pipe.front (a torch stream; torch.cuda.stream(pipe.front)) ...
        pipe.submit(slot, tag=shard)          # coder launch once `group` shards are waiting (or flush=True)
    pipe.drain()
"""
import ctypes as C

from . import host


class ShardPipeline:
    def __init__(self, batches, group=1, on_retire=None, sharded=False, trace=None, coder_streams=1):
        import torch
        self.torch = torch
        self.batches = list(batches)
        self.D = len(self.batches)
        self.G = max(1, int(group))
        if self.G > 1 and self.D < self.G:
            raise ValueError("a grouped pipeline needs at least `group` batches (2 * group when launches are left to fill up)")
        self.L = host.lib()
        L = self.L
        L.scalce_pipeline_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.scalce_pipeline_destroy.argtypes = [C.c_void_p]
        L.scalce_pipeline_destroy.restype = None
        L.scalce_pipeline_error.argtypes = [C.c_void_p]
        L.scalce_pipeline_error.restype = C.c_char_p
        for f in (L.scalce_pipeline_front_stream, L.scalce_pipeline_coder_stream):
            f.restype = C.c_void_p
        L.scalce_pipeline_front_stream.argtypes = [C.c_void_p]
        L.scalce_pipeline_coder_stream.argtypes = [C.c_void_p, C.c_int]
        L.scalce_pipeline_acquire.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.scalce_pipeline_submit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.scalce_pipeline_retire.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.scalce_pipeline_drain.argtypes = [C.c_void_p]
        L.scalce_pipeline_flush_last.argtypes = [C.c_void_p]
        arr = (C.c_void_p * self.D)(*[b.h for b in self.batches])
        h = C.c_void_p()
        rc = L.scalce_pipeline_create(arr, self.D, self.G, max(1, int(coder_streams)), 1 if sharded else 0, C.byref(h))
        self.h = h
        self._check(rc)
        # the library's streams as torch streams (callers enqueue torch work and their own C ABI calls on them)
        self.front = torch.cuda.ExternalStream(L.scalce_pipeline_front_stream(self.h))
        self.coders = [torch.cuda.ExternalStream(L.scalce_pipeline_coder_stream(self.h, i)) for i in range(max(1, int(coder_streams)))]
        self.coder = self.coders[0]
        self.on_retire = on_retire
        self.sharded = sharded
        self.trace = trace
        self._tag = [None] * self.D
        self._next = 0

    def _check(self, rc):
        if rc:
            msg = self.L.scalce_pipeline_error(self.h).decode() if self.h else "scalce_pipeline_create"
            ctx = self.batches[0].ctx
            raise RuntimeError(f"scalce pipeline: {msg} (rc {rc}): {self.L.scalce_last_error(ctx.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.scalce_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- slots ---------------------------------------------------------------------------------------------
    def _retired(self, slot):
        if self.trace:
            self.trace(f"slot {slot}: coder event reached")
        if self.on_retire:
            self.on_retire(slot, self.batches[slot], self._tag[slot])
        self._tag[slot] = None

    def acquire(self):
        """Next batch in round-robin order, free to be overwritten (its previous shard retired)."""
        slot, had = C.c_int(), C.c_int()
        self._check(self.L.scalce_pipeline_acquire(self.h, C.byref(slot), C.byref(had)))
        self._next = (slot.value + 1) % self.D
        if had.value:
            self._retired(slot.value)
        return slot.value, self.batches[slot.value]

    def retire(self, slot):
        had = C.c_int()
        self._check(self.L.scalce_pipeline_retire(self.h, slot, C.byref(had)))
        if had.value:
            self._retired(slot)

    # -- coder launches ------------------------------------------------------------------------------------
    def submit(self, slot, tag=None, flush=False):
        """The front stages of `slot` are enqueued on `self.front`: launch the coder now or with the next shards.
        flush: True / 1 = launch what is pending now and nothing runs beside it (the end of a run, a wave that fills every
        slot): shaped for its own latency; 2 = launch now, front stages of further shards follow beside it."""
        self._tag[slot] = tag
        launched = C.c_int()
        self._check(self.L.scalce_pipeline_submit(self.h, slot, 1 if flush is True else int(flush), C.byref(launched)))
        if launched.value and self.trace:
            self.trace("coder launched")

    def drain(self):
        """What is pending goes out as the last launch of the run, every slot is waited for and collected (on_retire in
        slot order).  With nobody to tell, scalce_pipeline_drain does the same in one call."""
        if not (self.on_retire or self.trace):
            self._check(self.L.scalce_pipeline_drain(self.h))
            return
        self._check(self.L.scalce_pipeline_flush_last(self.h))
        for slot in range(self.D):
            self.retire(slot)
