#!/usr/bin/env python3
"""Scale check of the paired-end path (BASELINE configs[2] shape, 150 bp, -r): one shard of N pairs through the whole hot
path, decoder round trip of both mates' quality streams on the device, timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scalce_amd import host, synth_gpu, format as fmt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
L = 150
dev = torch.device("cuda", 0)
blob = open(os.path.join(ROOT, "tests", "golden", "patterns.bin"), "rb").read()
ctx = host.Context(0, patterns_bin=blob)
t1 = synth_gpu.fastq_on_device(n, L, dev, seed=41, first_index=0)
t2 = synth_gpu.fastq_on_device(n, L, dev, seed=42, first_index=0)
off, vals, Ls = fmt.sample_qmap(t1[: 100000 * (2 * L + 20)].cpu().numpy().tobytes())
assert Ls == L
b = host.Batch(ctx, L, max_reads=n + 8, max_text=max(t1.numel(), t2.numel()) + 64, paired=True, read_len2=L,
               qmap=[(off, vals), (off, vals)])
torch.cuda.synchronize()
for it in range(2):
    t0 = time.perf_counter()
    b.compress(t1.data_ptr(), t1.numel(), t2.data_ptr(), t2.numel())
    b.finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
nin = t1.numel() + t2.numel()
print(f"{n} pairs x {L} bp: {dt * 1e3:.0f} ms, {nin / dt / 1e9:.1f} GB/s of FASTQ ({nin / 1e9:.1f} GB in), stats {b.stats()}")
assert b.n_reads == n
for m in (0, 1):
    nsym = n * L
    out = torch.zeros(nsym, dtype=torch.uint8, device=dev)
    p, nbytes = b.output_ptr(host.OUT_QUAL, m)
    ctx.ac_decode(b.output(host.OUT_TABLE, m, np.uint32), p, nbytes, nsym, out.data_ptr())
    qp, qn = b.output_ptr(host.OUT_QSTREAM, m)
    want = torch.empty(nsym, dtype=torch.uint8, device=dev)
    ctx.copy_d2d(want.data_ptr(), qp, qn)
    torch.cuda.synchronize()
    assert torch.equal(out, want), f"mate {m + 1}: decoded stream differs"
    print(f"mate {m + 1}: {nbytes / 1e9:.2f} GB coded, decoder round trip ok; reads payload {b.output_ptr(host.OUT_READS, m)[1] / 1e9:.2f} GB")
