#!/usr/bin/env python3
"""Diagnostic: one coder launch alone over a synthetic symbol stream (N(28, 8) clipped to 0..41, order-2 table counted
from the stream itself), for each SCALCE_AC_BLOCKS_PER_WG given on the command line.
usage: tools/coder_alone.py [blocks=512] [variants=1,8,64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scalce_amd import host

nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 512
variants = (sys.argv[2] if len(sys.argv) > 2 else "1,8,64").split(",")
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
for v in variants:
    os.environ["SCALCE_AC_BLOCKS_PER_WG"] = v
    b = host.Batch(ctx, 100, max_reads=1024, max_text=1 << 20)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.entropy_stream(0, table.data_ptr(), sym.data_ptr(), nsym)
        b.finish()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    nb = b.output_ptr(host.OUT_QUAL, 0)[1]
    print(f"blocks per wave {v:>3}: {dt * 1e3:8.1f} ms for {nblk} blocks ({dt * 1e9 / (10 * 1024 * 1024):.1f} ns per symbol of a block), {nb} bytes out", flush=True)
    b.close()
