"""Shards in flight: the host-side scheduling loop around the C ABI's split entropy stage.

The reference keeps its cores busy by handing `-T` blocks per batch to coder threads while the reader goes on
(arithmetic.cpp:349-357, compress.cpp:781-786).  The device-side counterpart: the arithmetic coder is a long kernel
of one wavefront per 10 MiB block (or per four / eight blocks) that leaves the memory system and most lanes idle, so
the front stages (ingest .. emit, and in a sharded run their collectives) of the NEXT shards run beside it on another
stream.  What the measurements on MI355X fixed:

* one `front` stream and `coder_streams` coder streams (bench.py: three, and GPU_MAX_HW_QUEUES=8 so that no two of them
  share a hardware queue): HIP maps streams onto a handful of hardware queues, and any read-back that lands in a queue
  behind a 0.6 s coder kernel waits for all of it;
* shards are retired on an EVENT recorded behind their coder launch and read back over the front stream;
* a launch takes `group` shards (scalce_batch_entropy_begin_group).  The one-block-per-lane coder holds ~10 CUs per shard
  for ~0.6 s whatever the launch holds, so consecutive launches rotate over the coder streams and run side by side
  (rounds 1-2, a wavefront per 1-8 blocks: one launch at a time on one stream);
* `slots >= 2 * group` batches, so that the front stages of one group run while the previous groups are coded.

Usage:
    pipe = ShardPipeline(batches, group=3)
    for shard in shards:
        slot, batch = pipe.acquire()          # waits for the batch's previous shard, calls on_retire for it
        ... front stages of `shard` into `batch` on pipe.front (a torch stream; torch.cuda.stream(pipe.front)) ...
        pipe.submit(slot, tag=shard)          # coder launch once `group` shards are waiting (or flush=True)
    pipe.drain()
"""
from . import host


class ShardPipeline:
    def __init__(self, batches, group=1, on_retire=None, sharded=False, trace=None, coder_streams=1):
        import torch
        self.torch = torch
        self.batches = list(batches)
        self.D = len(self.batches)
        self.G = max(1, int(group))
        if self.G > 1 and self.D < 2 * self.G:
            raise ValueError("a grouped pipeline needs at least 2 * group batches")
        self.front = torch.cuda.Stream()
        # coder launches rotate over `coder_streams` streams: the one-block-per-lane coder (ac_encode_lanes_k) takes a
        # fraction of a CU per 64 blocks for ~0.5 s whatever the size of the launch, so several launches run side by side
        self.coders = [torch.cuda.Stream() for _ in range(max(1, int(coder_streams)))]
        self.coder = self.coders[0]
        self.tail_streams = [torch.cuda.Stream() for _ in range(3)]  # the halves of a run's last group: never behind a running launch
        self._tail_launches = 0
        self._launches = 0
        self.on_retire = on_retire
        self.sharded = sharded          # shards arrive prepared (scalce_sharded_compress with SCALCE_SHARD_PREPARE_ONLY / _CODER_ASYNC)
        self.trace = trace
        self._busy = [None] * self.D    # event behind the slot's coder launch
        self._tag = [None] * self.D
        self._pending = []
        self._next = 0
        import os
        # launches of 1, 2, .. shards at the start of a run: measured (round 4) and left off -- the first slot comes back 80 ms
        # sooner, the front stages of the next shards run beside one more launch: 89.1 against 88.1 ms per shard at 20 steps
        self.ramp = not sharded and bool(os.environ.get("SCALCE_BENCH_RAMP"))
        self._run_launches = 0          # launches since the pipeline was last drained
        # the last group of a run in halves (submit): measured (round 4) and left off -- 88.4 against 86.4 ms per shard at 20 steps
        # (what the smaller, faster launches gain at the end their CUs take from the last front stages)
        self.tail = bool(os.environ.get("SCALCE_BENCH_TAIL"))

    # -- slots ---------------------------------------------------------------------------------------------
    def acquire(self):
        """Next batch in round-robin order, free to be overwritten (its previous shard retired)."""
        slot = self._next
        self._next = (self._next + 1) % self.D
        if slot in self._pending:  # caller never submitted enough shards to launch: flush before reuse
            self.flush()
        self.retire(slot)
        return slot, self.batches[slot]

    def retire(self, slot):
        ev = self._busy[slot]
        if ev is None:
            return
        ev.synchronize()
        if self.trace:
            self.trace(f"slot {slot}: coder event reached")
        self.batches[slot].finish(self.front.cuda_stream)  # sizes of the coded streams, device error word
        self._busy[slot] = None
        if self.on_retire:
            self.on_retire(slot, self.batches[slot], self._tag[slot])
        self._tag[slot] = None

    # -- coder launches ------------------------------------------------------------------------------------
    def submit(self, slot, tag=None, flush=False, remaining=None):
        """The front stages of `slot` are enqueued on `self.front`: launch the coder now or with the next shards.
        remaining: shards of the run still to come behind this one (None: unknown).  The END of a run is what the last
        launch takes after the last front stage -- ~0.6 s of an idle chip with one block per lane; with SCALCE_BENCH_TAIL=1 the
        last group goes out in halves (with 6 pending + to come: 3, then 2, then 1), the smaller ones with kernels that are
        sooner done (measured: no gain, off by default)."""
        self._tag[slot] = tag
        if self.G == 1:
            b = self.batches[slot]
            coder = self.coder
            if not self.sharded:  # a sharded caller has already enqueued the coder on self.coder (ent_stream)
                coder = self.coders[self._launches % len(self.coders)]
                self._launches += 1
                coder.wait_stream(self.front)
                b.entropy_begin(None, coder.cuda_stream)
            ev = self.torch.cuda.Event()
            ev.record(coder)
            self._busy[slot] = ev
            return
        self._pending.append(slot)
        # the first launches of a run are smaller (1, 2, .. shards): a launch takes ~0.6 s whatever it holds, and nothing is
        # coded -- no slot comes back -- until the first one has gone out
        target = min(self.G, self._run_launches + 1) if self.ramp else self.G
        tail = remaining is not None and self.tail and not self.sharded and remaining + len(self._pending) <= self.G
        if tail and not flush and remaining > 0:
            if len(self._pending) >= remaining:     # as many waiting as still to come: out they go
                self.flush(mode=3 if remaining <= 2 else 2)
            return
        if len(self._pending) >= target or flush:
            self.flush(last=flush, small=len(self._pending) < self.G and not flush)

    def flush(self, last=False, small=False, mode=None):
        """last: the caller has no further shards (the end of a run): the launch is picked for its own latency.
        small: a launch of fewer than `group` shards with more on their way: the kernel that holds the fewest CUs.
        mode: scalce_batch_entropy_begin_group_last's `last` argument given directly (tail launches)."""
        if not self._pending:
            return
        if mode is not None or (last and self._tail_launches):
            coder = self.tail_streams[self._tail_launches % len(self.tail_streams)]
            self._tail_launches += 1
        else:
            coder = self.coders[self._launches % len(self.coders)]
            self._launches += 1
        self._run_launches += 1
        host.entropy_begin_group([self.batches[sl] for sl in self._pending], self.front.cuda_stream, coder.cuda_stream,
                                 last=0 if self.sharded else (mode if mode is not None else 1 if last else 2 if small else 0))
        ev = self.torch.cuda.Event()
        ev.record(coder)
        for sl in self._pending:
            self._busy[sl] = ev
        self._pending = []
        if self.trace:
            self.trace("coder launched")

    def drain(self):
        self.flush(last=True)
        for slot in range(self.D):
            self.retire(slot)
        self._run_launches = 0
        self._tail_launches = 0
