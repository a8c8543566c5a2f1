"""One rank of the CPU rehearsal of the collectives (tests/test_shard_cpu.py): shared-memory transport in host mode."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    a = json.loads(sys.argv[1])
    from scalce_amd import host
    world, rank = a["world"], a["rank"]
    comm = host.Comm(-1, world, rank, shm_name=a["shm"], slot_bytes=1 << 20)
    L = host.lib()
    rng = np.random.default_rng(100 + rank)
    # all_gather
    mine = np.arange(1000, dtype=np.uint64) * (rank + 1)
    got = np.zeros(1000 * world, dtype=np.uint64)
    comm.all_gather(mine.ctypes.data, got.ctypes.data, mine.nbytes)
    assert (got.reshape(world, 1000) == np.arange(1000, dtype=np.uint64)[None, :] * (np.arange(world, dtype=np.uint64)[:, None] + 1)).all()
    # all_reduce (sum, u64)
    x = np.full(5000, rank + 1, dtype=np.uint64)
    comm.all_reduce_sum_u64(x.ctypes.data, len(x))
    assert (x == world * (world + 1) // 2).all()
    # all_to_all_v: rank r sends (r + 1) * (d + 1) * 100 bytes of value r * 16 + d to rank d
    send_counts = np.array([(rank + 1) * (d + 1) * 100 for d in range(world)], dtype=np.uint64)
    recv_counts = np.array([(s + 1) * (rank + 1) * 100 for s in range(world)], dtype=np.uint64)
    send = np.concatenate([np.full(int(send_counts[d]), rank * 16 + d, dtype=np.uint8) for d in range(world)])
    recv = np.zeros(int(recv_counts.sum()), dtype=np.uint8)
    comm._check(L.scalce_comm_all_to_all_v(comm.h, send.ctypes.data, send_counts.ctypes.data_as(C.POINTER(C.c_uint64)), recv.ctypes.data,
                                           recv_counts.ctypes.data_as(C.POINTER(C.c_uint64)), None))
    want = np.concatenate([np.full(int(recv_counts[s]), s * 16 + rank, dtype=np.uint8) for s in range(world)])
    assert (recv == want).all()
    # point to point, as a chain: rank r receives a running sum from r - 1, adds its own, hands it on (the tie-break's counts)
    L.scalce_comm_send.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    L.scalce_comm_recv.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    for rep in range(3):
        acc = np.zeros(777, dtype=np.uint64)
        if rank > 0:
            comm._check(L.scalce_comm_recv(comm.h, acc.ctypes.data, acc.nbytes, rank - 1, None))
        assert (acc == (rep + 1) * rank * (rank + 1) // 2).all()
        acc += (rep + 1) * (rank + 1)
        if rank + 1 < world:
            comm._check(L.scalce_comm_send(comm.h, acc.ctypes.data, acc.nbytes, rank + 1, None))
    comm.barrier()
    comm.close()
    print("rank", rank, "ok")


if __name__ == "__main__":
    main()
