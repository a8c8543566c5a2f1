"""-m gpu: shards in flight (scalce_amd/pipeline.py, the loop bench.py times) give every shard exactly the streams a
one-shot scalce_batch_compress gives it -- batches reused round-robin, several shards per coder launch, shards retired on
events while the next ones are already in their front stages."""
import numpy as np
import pytest

from scalce_amd import host, synth
from scalce_amd.pipeline import ShardPipeline

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("group,slots,shared", [(1, 1, False), (1, 2, False), (2, 4, False), (3, 6, False), (2, 4, True), (3, 6, True), (4, 8, True)])
def test_pipelined_shards_equal_one_shot_compress(group, slots, shared, patterns_blob):
    import torch
    from gpu_util import device_bytes
    ctx = host.Context(0, patterns_bin=patterns_blob)
    L = 100
    sizes = [40_000, 120_000, 5_000, 230_000, 60_000, 110_000, 90_000, 30_000]   # 1 .. 3 coder blocks, all different
    shards = []
    for i, n in enumerate(sizes):
        bases, quals = synth.reads_and_quals(n, L, seed=100 + i, dup_frac=0.1)
        fq = synth.fastq_bytes_fast(bases, quals, prefix=f"p{i}.")
        shards.append((device_bytes(fq), len(fq), n))
    want = []
    for t, nb, n in shards:   # one shard at a time, one-block-per-workgroup coder
        b = host.Batch(ctx, L, n + 8, nb + 64)
        b.compress(t.data_ptr(), nb)
        b.finish()
        want.append({w: b.output(w, 0).copy() for w in (host.OUT_QUAL, host.OUT_READS, host.OUT_NAMES)})
    nmax, bmax = max(sizes), max(s[1] for s in shards)
    # shared: ONE set of front-stage buffers for all slots (scalce_workspace): a shard's rows are overwritten by the next
    # shard's front stages while its own coder is still running
    ws = host.Workspace(ctx) if shared else None
    batches = [host.Batch(ctx, L, nmax + 8, bmax + 64, workspace=ws) for _ in range(slots)]
    got = {}

    def on_retire(slot, batch, tag):
        got[tag] = {w: batch.output(w, 0).copy() for w in (host.OUT_QUAL, host.OUT_READS, host.OUT_NAMES)}

    pipe = ShardPipeline(batches, group=group, on_retire=on_retire)
    torch.cuda.synchronize()
    for rep in range(2):   # every slot is reused several times
        for i, (t, nb, n) in enumerate(shards):
            slot, b = pipe.acquire()
            with torch.cuda.stream(pipe.front):
                b.front(t.data_ptr(), nb, None, 0, pipe.front.cuda_stream)
            pipe.submit(slot, tag=(rep, i), flush=(i + 1 == len(shards)))
    pipe.drain()
    assert len(got) == 2 * len(shards)
    for (rep, i), g in got.items():
        for w, name in ((host.OUT_QUAL, "qualities"), (host.OUT_READS, "reads"), (host.OUT_NAMES, "names")):
            assert len(g[w]) == len(want[i][w]) and (g[w] == want[i][w]).all(), f"shard {i} (pass {rep}): {name} differ"


def test_launches_in_waves_write_the_same_streams(patterns_blob, monkeypatch):
    """bench.py's launch plan for one GPU on its own: the pipeline's group is ALL its slots and the caller says when a launch
    goes out -- the remainder of the run first, beside the next front stages (flush 2), then waves that fill every slot and
    have the chip to themselves (flush 1: as few blocks per workgroup as all CUs allow).  Same streams as one shard at a
    time.  (SCALCE_AC_BLOCKS_PER_WG=64: the one-block-per-lane kernel whatever the size, so that its shape is what varies.)"""
    import torch
    from gpu_util import device_bytes
    ctx = host.Context(0, patterns_bin=patterns_blob)
    L = 100
    sizes = [230_000, 40_000, 120_000, 5_000, 60_000, 110_000, 90_000]
    shards = []
    for i, n in enumerate(sizes):
        bases, quals = synth.reads_and_quals(n, L, seed=300 + i, dup_frac=0.1)
        fq = synth.fastq_bytes_fast(bases, quals, prefix=f"w{i}.")
        shards.append((device_bytes(fq), len(fq), n))
    want = []
    for t, nb, n in shards:
        b = host.Batch(ctx, L, n + 8, nb + 64)
        b.compress(t.data_ptr(), nb)
        b.finish()
        want.append({w: b.output(w, 0).copy() for w in (host.OUT_QUAL, host.OUT_READS, host.OUT_NAMES)})
    monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", "64")
    slots = 5
    nmax, bmax = max(sizes), max(s[1] for s in shards)
    ws = host.Workspace(ctx)
    batches = [host.Batch(ctx, L, nmax + 8, bmax + 64, workspace=ws) for _ in range(slots)]
    for b in batches:
        b.set_code_in_place(True)
    got = {}

    def on_retire(slot, batch, tag):
        got[tag] = {w: batch.output(w, 0).copy() for w in (host.OUT_QUAL, host.OUT_READS, host.OUT_NAMES)}

    pipe = ShardPipeline(batches, group=slots, on_retire=on_retire, coder_streams=2)
    torch.cuda.synchronize()
    plan = {1: 2, 6: 1}   # 7 shards on 5 slots: 2 + 5
    for i, (t, nb, n) in enumerate(shards):
        slot, b = pipe.acquire()
        with torch.cuda.stream(pipe.front):
            b.front(t.data_ptr(), nb, None, 0, pipe.front.cuda_stream)
        pipe.submit(slot, tag=i, flush=plan.get(i, 0))
    pipe.drain()
    assert len(got) == len(shards)
    for i, g in got.items():
        for w, name in ((host.OUT_QUAL, "qualities"), (host.OUT_READS, "reads"), (host.OUT_NAMES, "names")):
            assert len(g[w]) == len(want[i][w]) and (g[w] == want[i][w]).all(), f"shard {i}: {name} differ"


@pytest.mark.parametrize("bpw,overflow", [("", False), ("64", False), ("8", False), ("64", True), ("", True)])
def test_coding_in_place_writes_the_same_streams(bpw, overflow, patterns_blob, monkeypatch):
    """scalce_batch_set_code_in_place: the coder's blocks go over the symbols they were coded from (no block buffers: 3.2 GB less
    per 50 M-read shard in flight) -- the same bytes through every kernel a grouped launch takes (four / eight blocks per chain
    wave, one block per lane).  overflow: SCALCE_AC_INPLACE_TEST makes the kernels' bound so tight that a block's output catches
    up with its input: its symbols are gone, the shard is run again from its text (arithmetic.cpp:85-169 knows no such case:
    its output buffer is a buffer of its own)."""
    import torch
    from gpu_util import device_bytes
    if bpw:
        monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", bpw)
    if overflow:
        monkeypatch.setenv("SCALCE_AC_INPLACE_TEST", "1")
    ctx = host.Context(0, patterns_bin=patterns_blob)
    L = 100
    sizes = [230_000, 120_000, 5_000, 40_000, 330_000, 110_000]   # 1 .. 4 coder blocks, the last one short or tiny
    shards, want = [], []
    for i, n in enumerate(sizes):
        bases, quals = synth.reads_and_quals(n, L, seed=300 + i, dup_frac=0.1)
        fq = synth.fastq_bytes_fast(bases, quals, prefix=f"q{i}.")
        shards.append((device_bytes(fq), len(fq), n))
    monkeypatch.delenv("SCALCE_AC_BLOCKS_PER_WG", raising=False)
    for t, nb, n in shards:   # one shard at a time, block buffers of its own
        b = host.Batch(ctx, L, n + 8, nb + 64)
        b.compress(t.data_ptr(), nb)
        b.finish()
        want.append({w: b.output(w, 0).copy() for w in (host.OUT_QUAL, host.OUT_READS, host.OUT_NAMES)})
        b.close()
    if bpw:
        monkeypatch.setenv("SCALCE_AC_BLOCKS_PER_WG", bpw)
    nmax, bmax = max(sizes), max(s[1] for s in shards)
    ws = host.Workspace(ctx)
    batches = [host.Batch(ctx, L, nmax + 8, bmax + 64, workspace=ws) for _ in range(6)]
    for b in batches:
        b.set_frame_on_demand(True)
        b.set_code_in_place(True)
    got = {}

    def on_retire(slot, batch, tag):
        got[tag] = {w: batch.output(w, 0).copy() for w in (host.OUT_QUAL, host.OUT_READS, host.OUT_NAMES)}

    pipe = ShardPipeline(batches, group=3, on_retire=on_retire)
    torch.cuda.synchronize()
    for rep in range(2):
        for i, (t, nb, n) in enumerate(shards):
            slot, b = pipe.acquire()
            with torch.cuda.stream(pipe.front):
                b.front(t.data_ptr(), nb, None, 0, pipe.front.cuda_stream)
            pipe.submit(slot, tag=(rep, i), flush=(i + 1 == len(shards)))
    pipe.drain()
    assert len(got) == 2 * len(shards)
    for (rep, i), g in got.items():
        for w, name in ((host.OUT_QUAL, "qualities"), (host.OUT_READS, "reads"), (host.OUT_NAMES, "names")):
            assert len(g[w]) == len(want[i][w]) and (g[w] == want[i][w]).all(), f"shard {i} (pass {rep}): {name} differ"
    reruns = sum(b.reruns for b in batches)
    assert (reruns > 0) == overflow, f"{reruns} shards were run again from their text"
