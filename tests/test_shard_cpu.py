"""CPU: the host-side half of sharded runs -- the plan math of scalce_amd/csrc/sharded.cpp against a plain restatement,
and the collectives' semantics across real processes (shared-memory transport in host mode, world 2 and 3)."""
import ctypes as C
import json
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

from scalce_amd import host

HERE = os.path.dirname(os.path.abspath(__file__))
AC_BLOCK = 10 * 1024 * 1024


def ref_block_plan(Cm, L, rank):
    """Plain restatement: the run-wide reordered stream is bucket-major, rank-major inside a bucket; rank d gets the
    blocks [d * nblk // world, (d + 1) * nblk // world)."""
    Cm = np.asarray(Cm, dtype=np.int64)
    world, _ = Cm.shape
    Cg = Cm.sum(axis=0)
    total = int(Cg.sum()) * L
    nblk = -(-total // AC_BLOCK)
    lo = [min(total, (d * nblk // world) * AC_BLOCK) for d in range(world + 1)]
    base = (np.cumsum(Cg) - Cg) * L
    before = np.cumsum(Cm, axis=0) - Cm
    g0 = base[None, :] + before * L
    ln = Cm * L

    def below(r, X):
        return int(np.clip(X - g0[r], 0, ln[r]).sum())

    send = [below(rank, lo[d + 1]) - below(rank, lo[d]) for d in range(world)]
    recv = [below(s, lo[rank + 1]) - below(s, lo[rank]) for s in range(world)]
    psrc, pdst, at = [], [], 0
    for s in range(world):
        a = np.maximum(g0[s], lo[rank])
        b = np.minimum(g0[s] + ln[s], lo[rank + 1])
        for k in np.flatnonzero(b > a):
            psrc.append(at)
            pdst.append(int(a[k]) - lo[rank])
            at += int(b[k] - a[k])
    return send, recv, lo[rank], lo[rank + 1], psrc, pdst


@pytest.mark.parametrize("world,nb1,L,scale", [(2, 50, 100, 40000), (3, 200, 150, 9000), (8, 1000, 100, 3000), (4, 30, 36, 50)])
def test_block_plan_matches_restatement(world, nb1, L, scale):
    rng = np.random.default_rng(world * 1000 + nb1)
    Cm = rng.integers(0, scale, size=(world, nb1)).astype(np.uint64)
    Cm[rng.random(size=Cm.shape) < 0.2] = 0
    lib = host.lib()
    u64p = C.POINTER(C.c_uint64)
    lib.scalce_shard_plan_blocks.argtypes = [C.c_int, C.c_int, C.c_uint32, u64p, C.c_uint64, u64p, u64p, u64p, u64p, u64p, u64p, u64p]
    sent = np.zeros((world, world), dtype=np.int64)
    for rank in range(world):
        send = np.zeros(world, dtype=np.uint64)
        recv = np.zeros(world, dtype=np.uint64)
        ps = np.zeros(world * nb1, dtype=np.uint64)
        pd = np.zeros(world * nb1, dtype=np.uint64)
        lo, hi, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
        flat = np.ascontiguousarray(Cm.reshape(-1))
        rc = lib.scalce_shard_plan_blocks(world, rank, nb1, flat.ctypes.data_as(u64p), L, send.ctypes.data_as(u64p), recv.ctypes.data_as(u64p),
                                          C.byref(lo), C.byref(hi), ps.ctypes.data_as(u64p), pd.ctypes.data_as(u64p), C.byref(n))
        assert rc == 0
        rs, rr, rlo, rhi, rps, rpd = ref_block_plan(Cm, L, rank)
        assert list(send) == rs and list(recv) == rr and (lo.value, hi.value) == (rlo, rhi)
        assert list(ps[: n.value]) == rps and list(pd[: n.value]) == rpd
        assert lo.value % AC_BLOCK == 0 and int(recv.sum()) == hi.value - lo.value
        sent[rank] = send
    # what d receives from s is what s sends to d
    for d in range(world):
        _, rr, *_ = ref_block_plan(Cm, L, d)
        assert list(sent[:, d]) == rr


def test_boundaries_move_to_the_nearest_cut():
    lib = host.lib()
    u64p = C.POINTER(C.c_uint64)
    lib.scalce_shard_plan_boundaries.argtypes = [C.c_int, u64p, u64p, C.c_uint64, u64p]

    def plan(g, cuts):
        g = np.array(g, dtype=np.uint64)
        c = np.array(sorted(cuts), dtype=np.uint64)
        out = np.zeros(len(g), dtype=np.uint64)
        rc = lib.scalce_shard_plan_boundaries(len(g) - 1, g.ctypes.data_as(u64p), c.ctypes.data_as(u64p), len(c), out.ctypes.data_as(u64p))
        return rc, list(map(int, out))

    assert plan([0, 100, 200, 300], [40, 90, 160, 210, 290]) == (0, [0, 90, 210, 300])
    assert plan([0, 100, 200], [100]) == (0, [0, 100, 200])                    # already on a cut
    assert plan([0, 100, 200], [60, 140]) == (0, [0, 60, 200])                 # a tie goes to the earlier cut
    assert plan([0, 10, 20, 30], [20]) == (0, [0, 20, 20, 30])                 # rank 1 ends up without records
    assert plan([0, 100], []) == (0, [0, 100])                                 # one rank: nothing to move
    assert plan([0, 100, 200], []) == (0, [0, 200, 200])                       # no cut at all: one chunk, all of it to rank 0
    assert plan([0, 10, 20, 30, 40], []) == (0, [0, 40, 40, 40, 40])
    assert plan([0, 10, 20, 30, 40], [35]) == (0, [0, 35, 35, 35, 40])         # one cut for three boundaries: two ranks end up empty


@pytest.mark.parametrize("world", [2, 3])
def test_collectives_across_processes(world):
    shm = "/scalce_cpu_" + uuid.uuid4().hex[:12]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "comm_worker.py"), json.dumps(dict(world=world, rank=r, shm=shm))],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=120)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0 and f"rank {r} ok" in logs[r], logs[r][-2000:]
