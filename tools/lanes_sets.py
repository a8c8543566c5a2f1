#!/usr/bin/env python3
"""Diagnostic: ac_encode_lanes_k alone, one launch of `blocks` blocks, for every (sets of four waves per workgroup, lanes
in use per set) given: how long the launch takes, how many CUs it holds, and that every variant writes the same bytes.
usage: tools/lanes_sets.py [blocks=1431] [variants=1:48,2:48,2:40,2:32]   (a variant: sets:lanes[:pairing[:priority of the light waves]])"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scalce_amd import host

nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 1431
variants = (sys.argv[2] if len(sys.argv) > 2 else "1:48,2:48,2:40,2:32").split(",")
dev = torch.device("cuda", 0)
ctx = host.Context(0, patterns_bin=open(os.path.join(ROOT, "tests", "golden", "patterns.bin"), "rb").read())
nsym = nblk * 10 * 1024 * 1024
g = torch.Generator(device=dev); g.manual_seed(7)
sym = torch.empty(nsym + 64, dtype=torch.uint8, device=dev)
step = 1 << 28
for a in range(0, nsym, step):
    k = min(step, nsym - a)
    sym[a:a + k] = torch.clamp(torch.round(torch.randn(k, device=dev, generator=g) * 8 + 28), 0, 41).to(torch.uint8)
head = sym[: 1 << 24].to(torch.int64)
idx = (head[:-2] * 80 + head[1:-1]) * 80 + head[2:]
table = (torch.bincount(idx, minlength=512000) + 1).to(torch.int32)
torch.cuda.synchronize()
os.environ["SCALCE_AC_BLOCKS_PER_WG"] = "64"
ref = None
for v in variants:
    f = v.split(":")
    sets, lanes = f[0], f[1]
    os.environ["SCALCE_AC_SETS"] = sets
    os.environ["SCALCE_AC_LANES_USED"] = lanes
    os.environ["SCALCE_AC_PAIRING"] = f[2] if len(f) > 2 else "0"
    os.environ["SCALCE_AC_LIGHT_PRIO"] = f[3] if len(f) > 3 else "2"
    b = host.Batch(ctx, 100, max_reads=1024, max_text=1 << 20)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.entropy_stream(0, table.data_ptr(), sym.data_ptr(), nsym)
        b.finish()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out = b.output(host.OUT_QUAL, 0)
    dig = hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest()[:16]
    cus = -(-nblk // (int(sets) * int(lanes)))
    print(f"sets {sets} lanes {lanes} ({v}): {dt * 1e3:8.1f} ms for {nblk} blocks on {cus} CUs = {dt * cus:.2f} CU-s "
          f"({dt * 1e9 / (10 * 1024 * 1024):.1f} ns per symbol of a block), {len(out)} bytes out, sha {dig}", flush=True)
    if ref is None: ref = dig
    assert dig == ref, "coded bytes differ between variants"
    b.close()
