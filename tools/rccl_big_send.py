#!/usr/bin/env python3
"""Where does a large RCCL send / receive stop?  (round 4 found that a 5 GB ncclSend / ncclRecv of one rank to itself did not
arrive whole; comm.cpp sends in pieces of 1 GiB since.)  One rank, the library's own communicator, its piece size raised so
far that every message goes out as ONE ncclSend / ncclRecv pair: for several sizes around 2^32 bytes, how many leading
bytes of the message arrived and where the first wrong byte is.
usage (GPU box):  python tools/rccl_big_send.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from scalce_amd import host  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    comm = host.Comm(0, 1, 0, unique_id=host.Comm.unique_id())
    comm.L.scalce_comm_set_piece_bytes.argtypes = [C.c_void_p, C.c_uint64]
    comm.L.scalce_comm_set_piece_bytes.restype = None
    comm.L.scalce_comm_set_piece_bytes(comm.h, 1 << 40)   # every message as ONE ncclSend / ncclRecv pair
    G = 1 << 30
    for n in (G, 2 * G - 4096, 2 * G, 2 * G + (1 << 20), 3 * G, 4 * G - 4096, 4 * G, 4 * G + (1 << 20), 5 * G + 12345):
        src = (torch.arange(n, device=dev, dtype=torch.int64) * 2654435761 >> 7).to(torch.uint8)
        dst = torch.zeros(n, dtype=torch.uint8, device=dev)
        comm.all_to_all_v(src.data_ptr(), [n], dst.data_ptr(), [n])
        torch.cuda.synchronize()
        nbad, first, last = 0, -1, -1
        for a in range(0, n, 1 << 28):       # (a nonzero() over billions of elements overflows torch's index arithmetic)
            neq = src[a:a + (1 << 28)] != dst[a:a + (1 << 28)]
            k = int(neq.sum())
            if k:
                nbad += k
                idx = neq.to(torch.uint8)
                if first < 0:
                    first = a + int(idx.argmax())
                last = a + idx.numel() - 1 - int(idx.flip(0).argmax())
            del neq
        print(f"one ncclSend/ncclRecv of {n} bytes ({n / G:.3f} GiB): {nbad} wrong bytes, first at {first}, last at {last}"
              + (f" (first = 2^31 {first - (1 << 31):+d} = 2^32 {first - (1 << 32):+d})" if first >= 0 else ""), flush=True)
        del src, dst
        torch.cuda.empty_cache()
    comm.close()


if __name__ == "__main__":
    main()
