"""-m gpu: a run sharded over virtual ranks (threads, one GPU) produces ONE archive that equals the
single-shard run with one spill chunk per shard -- tokens incl. the run-wide tie-break, record order, names,
quality table and arithmetic-coder blocks cut on the run-wide stream -- and both equal the oracle."""
import os
import struct

import numpy as np
import pytest

import oraclelib as O
from scalce_amd import dist, host, synth

pytestmark = pytest.mark.gpu


def _shard_text(bases, quals, bounds, r):
    a, b = bounds[r], bounds[r + 1]
    out = []
    for i in range(a, b):
        out.append(b"@s.%d\n" % i + bases[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("world,n,L,deferred", [(3, 9000, 100, False), (2, 230000, 100, False), (4, 5000, 36, False),
                                                 (2, 230000, 100, True), (3, 9000, 100, True), (2, 230000, 100, "group")])
def test_sharded_equals_single_with_one_chunk_per_shard(world, n, L, deferred, patterns_blob):
    import torch
    from gpu_util import device_bytes
    fourmers = None
    if L == 36:  # every position a tie: stresses the cross-shard fixed point
        import itertools
        fourmers = ("\n".join("".join(x) for x in itertools.product("ACGT", repeat=4)) + "\n").encode()
        ctx, trie = host.Context(0, patterns_text=fourmers), O.Trie(text=fourmers)
    else:
        ctx, trie = host.Context(0, patterns_bin=patterns_blob), O.Trie(blob=patterns_blob)
    bases, quals = synth.reads_and_quals(n, L, seed=91, dup_frac=0.15, n_frac=0.002)
    cuts = np.linspace(0, n, world + 1).astype(int)
    cuts[1:-1] += np.array([7, -13, 5][: world - 1])  # uneven shards
    texts = [_shard_text(bases, quals, cuts, r) for r in range(world)]
    whole = b"".join(texts)

    # single shard, explicit chunks at the shard boundaries
    tw = device_bytes(whole)
    single = host.Batch(ctx, L, n + 8, len(whole) + 64)
    single.set_chunks(cuts[:-1])
    single.compress(tw.data_ptr(), len(whole))
    single.finish()
    assert single.stats()["chunks"] == world

    # oracle with the same chunks
    pat, end = trie.tokenize(bases)
    chunk = np.zeros(n, dtype=np.int32)
    for r in range(world):
        chunk[cuts[r]:cuts[r + 1]] = r
    perm = trie.order(bases, pat, end, chunk)
    tok = single.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)
    assert (tok[:, 0] == pat).all() and (tok[:, 1] == end).all()
    assert (single.output(host.OUT_PERM, 0, np.uint32) == perm).all()
    qp, f4 = O.quality_stream(quals, bases, 33, np.arange(128))
    want_q = O.AcStat(O.ac_scale(f4, 1)).encode_stream(qp[perm].reshape(-1))
    assert (single.output(host.OUT_QUAL, 0) == want_q).all()

    # sharded over virtual ranks
    dtexts = [device_bytes(t) for t in texts]
    batches = [host.Batch(ctx, L, (cuts[r + 1] - cuts[r]) + 8, len(texts[r]) + 64) for r in range(world)]

    def body(comm):
        r = comm.rank
        if not deferred:
            return dist.compress_shard(comm, ctx, batches[r], dtexts[r].data_ptr(), len(texts[r]))
        if deferred == "group":
            # the way bench.py --group does it: the shard only hands its range of the run-wide stream to the batch,
            # the coder launch comes later and may hold other shards' blocks as well (here: this rank's shard alone)
            res = dist.compress_shard(comm, ctx, batches[r], dtexts[r].data_ptr(), len(texts[r]), prepare_only=True)
            host.entropy_begin_group([batches[r]])
            batches[r].finish()
            return res
        # the way bench.py keeps shards in flight: front stages on one stream, the coder only enqueued on another
        front, ent = torch.cuda.Stream(), torch.cuda.Stream()
        front.wait_stream(torch.cuda.default_stream())
        with torch.cuda.stream(front):
            res = dist.compress_shard(comm, ctx, batches[r], dtexts[r].data_ptr(), len(texts[r]),
                                      stream=front.cuda_stream, ent_stream=ent)
        batches[r].finish(ent.cuda_stream)
        return res

    results = dist.run_threads(world, body)
    toks = np.concatenate([b.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2) for b in batches])
    bad = np.flatnonzero((toks[:, 0] != pat) | (toks[:, 1] != end))
    assert len(bad) == 0, f"{len(bad)} token mismatches, first {bad[:5]} sweeps {[r.sweeps for r in results]}"
    got = dist.assemble(ctx, results, batches, L)
    assert got["reads"] == single.output(host.OUT_READS, 0).tobytes()
    assert got["names"] == single.output(host.OUT_NAMES, 0).tobytes()
    assert (got["table"] == single.output(host.OUT_TABLE, 0, np.uint32)).all()
    assert got["qual"] == single.output(host.OUT_QUAL, 0).tobytes(), "AC blocks on the run-wide stream differ"
    print("sweeps", [r.sweeps for r in results], "blocks", [len(b.output(host.OUT_QUAL, 0)) for b in batches])
