#!/usr/bin/env python3
"""A/B timing of two builds of the scalce binary on the same box and the same input file (tmpfs), alternating runs.
   python tools/cli_ab.py READS DIR BIN_A BIN_B [compress|decompress]"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from scalce_amd import synth_gpu  # noqa: E402

n = int(sys.argv[1])
d = sys.argv[2]
bins = sys.argv[3:5]
what = sys.argv[5] if len(sys.argv) > 5 else "compress"
os.makedirs(d, exist_ok=True)
fq = os.path.join(d, "in_1.fq")
text = synth_gpu.fastq_on_device(n, 100, torch.device("cuda", 0), seed=7, first_index=0)
text.cpu().numpy().tofile(fq)
size = os.path.getsize(fq)
del text
torch.cuda.empty_cache()
open(fq, "rb").read()  # (a file another process has just written reads slowly the first time)
pbin = os.path.join(ROOT, "tests", "golden", "patterns.bin")
if what == "decompress":
    subprocess.run([os.path.join(ROOT, "scalce_amd", "bin", bins[0]), "-c", "no", "-o", os.path.join(d, "arc"), fq, "--patterns-bin", pbin], check=True, capture_output=True)
for rep in range(3):
    for b in bins:
        exe = os.path.join(ROOT, "scalce_amd", "bin", b)
        out = os.path.join(d, "out_" + b)
        for f in os.listdir(d):
            if f.startswith("out_"):
                os.remove(os.path.join(d, f))
        cmd = [exe, "-c", "no", "-o", out, fq, "--patterns-bin", pbin] if what == "compress" else [exe, "-d", "-o", out, os.path.join(d, "arc_1.scalcen"), "--patterns-bin", pbin]
        t = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True)
        dt = time.perf_counter() - t
        assert r.returncode == 0, r.stderr[-1500:]
        m = re.search(r"Time elapsed: (.*)", r.stderr)
        print(f"{b:12s} {what} {dt:.2f} s wall = {size / dt / 1e6:.0f} MB/s  [{m.group(1) if m else ''}]", flush=True)
