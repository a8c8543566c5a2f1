"""CPU, gloo, world_size 2-3: the collective plumbing of sharded runs (scalce_amd/dist.py) without a GPU.
The per-shard compute is stood in by the oracle here (tests only): what is under test is the exchange logic --
block-range plan, all-to-all splits, piece placement, boundary trigrams, the cross-rank fixed point."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

import oraclelib as O
from scalce_amd import dist, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(dist.TorchComm())
    finally:
        tdist.destroy_process_group()


def spawn(world, fn):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _quality_exchange(comm):
    """Each rank holds a bucket-major local stream; after plan + all_to_all + piece placement every rank must hold
    exactly its block range of the run-wide stream."""
    rng = np.random.default_rng(100)
    world, nb1, L = comm.world, 57, 100
    C = rng.integers(0, 2500, size=(world, nb1))
    C[:, rng.integers(0, nb1, 9)] = 0
    # symbol value encodes (rank, bucket, index) so that any misplacement shows
    def local_stream(r):
        parts = [((np.arange(C[r][b] * L) * 7 + 13 * b + 101 * r) % 251).astype(np.uint8) for b in range(nb1)]
        return np.concatenate(parts)
    streams = [local_stream(r) for r in range(world)]
    glob = np.concatenate([streams_r[(np.cumsum(C[r]) - C[r])[b] * L:(np.cumsum(C[r]))[b] * L]
                           for b in range(nb1) for r, streams_r in enumerate(streams)])
    plan = dist.stream_plan(C, L, comm.rank)
    got = comm.all_to_all(torch.from_numpy(streams[comm.rank]), plan["send"], plan["recv"]).numpy()
    mine = np.zeros(plan["hi"] - plan["lo"], dtype=np.uint8)
    ps, pd = plan["piece_src"].astype(np.int64), plan["piece_dst"].astype(np.int64)
    ends = np.concatenate([ps[1:], [len(got)]])
    for s, e, d in zip(ps, ends, pd):
        mine[d:d + (e - s)] = got[s:e]
    assert (mine == glob[plan["lo"]:plan["hi"]]).all()
    return plan["lo"], plan["hi"], plan["total"]


@pytest.mark.parametrize("world", [2, 3])
def test_block_range_all_to_all(world):
    out = spawn(world, _quality_exchange)
    total = out[0][2]
    assert out[0][0] == 0 and out[-1][1] == total
    for a, b in zip(out[:-1], out[1:]):
        assert a[1] == b[0] and a[1] % dist.AC_BLOCK == 0


def test_boundary_trigrams_complete_the_table():
    """Sum of per-shard tables (each starting without predecessors) + boundary trigrams == table of the whole."""
    bases, quals = synth.reads_and_quals(3000, 40, seed=3, n_frac=0.01)
    _, f_all = O.quality_stream(quals, bases, 33, np.arange(128))
    cuts = [0, 700, 700, 1901, 3000]  # includes an empty shard
    tot = np.zeros(512000, dtype=np.int64)
    edges, ns = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            qp, f = O.quality_stream(quals[a:b], bases[a:b], 33, np.arange(128))
            tot += f.astype(np.int64) - 1
            flat = qp.reshape(-1)
            edges.append([flat[0], flat[1], flat[-2], flat[-1]])
        else:
            edges.append([0, 0, 0, 0])
        ns.append((b - a) * 40)
    for k in dist.boundary_trigrams(np.array(edges), ns):
        tot[k] += 1
    assert (tot + 1 == f_all.astype(np.int64)).all()


def _fixed_point(comm):
    """Cross-rank Jacobi on the oracle's tie semantics: ranks re-decide their tie reads against prior counts
    obtained by all_gather until nobody changes; the result must equal the sequential tokenizer."""
    import itertools
    txt = ("\n".join("".join(x) for x in itertools.product("ACGT", repeat=3)) + "\n").encode()
    trie = O.Trie(text=txt)
    bases, _ = synth.reads_and_quals(600, 24, seed=17)
    want_pat, want_end = trie.tokenize(bases)
    cuts = np.linspace(0, 600, comm.world + 1).astype(int)
    a, b = cuts[comm.rank], cuts[comm.rank + 1]
    nb = trie.n_patterns
    # candidates per read: every distinct 3-mer in order of first appearance (all cores have length 3)
    pats = {trie.pattern(p).decode(): p for p in range(nb)}
    cands = []
    for r in range(a, b):
        s = bases[r].tobytes().decode()
        seen, lst = set(), []
        for i in range(len(s) - 2):
            p = pats[s[i:i + 3]]
            if p not in seen:
                seen.add(p)
                lst.append((p, i + 2))
        cands.append(lst)
    choice = [0] * (b - a)
    sweeps = 0
    while True:
        counts = np.zeros(nb, dtype=np.int64)
        for c, lst in zip(choice, cands):
            counts[lst[c][0]] += 1
        allc = comm.all_gather(torch.from_numpy(counts)).numpy()
        prior = allc[:comm.rank].sum(axis=0)
        run = prior.copy()
        changed = 0
        new_choice = []
        # Jacobi: decisions from the counts implied by the PREVIOUS choices of earlier local reads
        local_prefix = np.zeros(nb, dtype=np.int64)
        for idx, lst in enumerate(cands):
            best, bestc = 0, -1
            for j, (p, _) in enumerate(lst):
                c = run[p] + local_prefix[p]
                if j == 0 or c > bestc:
                    best, bestc = j, c
            new_choice.append(best)
            changed |= best != choice[idx]
            local_prefix[lst[choice[idx]][0]] += 1
        choice = new_choice
        sweeps += 1
        if comm.all_reduce_max(int(changed)) == 0:
            break
    got_pat = np.array([cands[i][c][0] for i, c in enumerate(choice)])
    got_end = np.array([cands[i][c][1] + 1 for i, c in enumerate(choice)])
    assert (got_pat == want_pat[a:b]).all() and (got_end == want_end[a:b]).all()
    return sweeps


def test_cross_rank_fixed_point_matches_sequential():
    sweeps = spawn(2, _fixed_point)
    assert sweeps[0] == sweeps[1] and sweeps[0] >= 2
