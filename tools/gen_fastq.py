#!/usr/bin/env python3
"""Write a synthetic FASTQ file of any size: generated on the GPU in chunks (scalce_amd.synth_gpu: names @s.<i>, uniform
ACGT, qualities clip(round(N(30,8)),2,40)+33, bare '+'), appended to the file chunk by chunk.
usage: gen_fastq.py READS LENGTH PATH [SEED]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from scalce_amd import synth_gpu  # noqa: E402

n, L, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 20261003
dev = torch.device("cuda", 0)
step = 8_000_000
with open(path, "wb") as f:
    for a in range(0, n, step):
        m = min(step, n - a)
        t = synth_gpu.fastq_on_device(m, L, dev, seed=seed + a, first_index=a)
        f.write(memoryview(t.cpu().numpy()))
        del t
print(path, os.path.getsize(path), "bytes", flush=True)
