#!/usr/bin/env python3
"""Generate the synthetic core table used by every test and by bench.py.

The real SFU patterns.bin is not in the reference tree (.MISSING_LARGE_BLOBS), so parity is
defined relative to a table both sides are given.  Layout follows read_patterns
(/root/reference/reads.cpp:342-369): groups of [int16 len][int32 count] followed by `count`
little-endian integers of ceil(len/4) bytes, first base in the most significant 2 bits.
Recipe is SURVEY.md Appendix A: random.seed(786); (len,count) in (8,600),(10,3000),(12,12000);
distinct getrandbits(2*len) per group, each group sorted ascending -> 46 218 bytes, 15 600 cores.
"""
import random
import struct
import sys


def build(groups=((8, 600), (10, 3000), (12, 12000)), seed=786):
    random.seed(seed)
    out = bytearray()
    for ln, cnt in groups:
        vals = set()
        while len(vals) < cnt:
            vals.add(random.getrandbits(2 * ln))
        out += struct.pack("<hi", ln, cnt)
        nb = (ln + 3) // 4
        for v in sorted(vals):
            out += v.to_bytes(nb, "little")
    return bytes(out)


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "patterns.bin"
    blob = build()
    with open(path, "wb") as f:
        f.write(blob)
    print(path, len(blob), "bytes")
