#!/usr/bin/env python3
"""Time scalce_ac_decode on one 50 M x 100 bp shard: compact rows + LDS cache against the plain kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scalce_amd import host, synth_gpu, format as fmt

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000, 100
dev = torch.device("cuda", 0)
ctx = host.Context(0, patterns_bin=open(os.path.join(ROOT, "tests", "golden", "patterns.bin"), "rb").read())
text = synth_gpu.fastq_on_device(n, L, dev, seed=20261003, first_index=0)
nbytes = text.numel()
off, vals, Ls = fmt.sample_qmap(text[: 100000 * (2 * L + 20)].cpu().numpy().tobytes())
b = host.Batch(ctx, L, max_reads=n + 8, max_text=nbytes + 64, qmap=[(off, vals), (off, vals)])
b.compress(text.data_ptr(), nbytes, None, 0, 0); b.finish()
table = b.output(host.OUT_TABLE, 0, np.uint32)
p, nb = b.output_ptr(host.OUT_QUAL, 0)
nsym = n * L
out = torch.zeros(nsym, dtype=torch.uint8, device=dev)
qp, qn = b.output_ptr(host.OUT_QSTREAM, 0)
want = torch.empty(nsym, dtype=torch.uint8, device=dev)
ctx.copy_d2d(want.data_ptr(), qp, qn, 0)
variants = (("tight loop (round 4)", {}), ("plain kernel", {"SCALCE_AC_DECODE_WPB": "0"}))
if len(sys.argv) > 2:
    variants = variants[:int(sys.argv[2])]
for name, env in variants:
    os.environ.update(env)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.ac_decode(table, p, nb, nsym, out.data_ptr())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e3:.0f} ms for {nsym / 1e9:.1f} G symbols = {dt / (10 * 1024 * 1024) * 1e9:.0f} ns per symbol per block, "
          f"{nbytes / dt / 1e9:.2f} GB/s of FASTQ; equal: {bool(torch.equal(out, want))}", flush=True)
    out.zero_()
