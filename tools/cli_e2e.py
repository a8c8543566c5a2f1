#!/usr/bin/env python3
"""End-to-end timing of the scalce command line (file in, archive out, archive back to FASTQ) on a synthetic
file: what a user of the reference's CLI sees, PCIe and file I/O included."""
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from scalce_amd import synth_gpu  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp/scalce_e2e"
os.makedirs(d, exist_ok=True)
fq = os.path.join(d, "in_1.fq")
text = synth_gpu.fastq_on_device(n, 100, torch.device("cuda", 0), seed=7, first_index=0)
text.cpu().numpy().tofile(fq)
size = os.path.getsize(fq)
del text
torch.cuda.empty_cache()
cli = os.path.join(ROOT, "scalce_amd", "bin", "scalce")
pbin = os.path.join(ROOT, "tests", "golden", "patterns.bin")


def run(*a):
    t = time.perf_counter()
    r = subprocess.run([cli, *a], capture_output=True, text=True)
    dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-1500:]
    line = [x.strip() for x in r.stderr.splitlines() if "Time elapsed" in x or "Process:" in x]
    return dt, " | ".join(line)


def digest(path):  # order-independent digest of the 4-line records
    acc, cnt = 0, 0
    with open(path, "rb") as f:
        while True:
            rec = [f.readline() for _ in range(4)]
            if not rec[0]:
                break
            acc = (acc + zlib.crc32(b"".join(rec)) * 2654435761) & ((1 << 64) - 1)
            cnt += 1
    return cnt, acc


for cont in ("no", "gz"):
    out = os.path.join(d, "arc_" + cont)
    dt, line = run("-c", cont, "-o", out, fq, "--patterns-bin", pbin)
    asz = sum(os.path.getsize(f"{out}_1.scalce{e}") for e in "nrq")
    print(f"compress -c {cont}: {dt:.2f} s wall = {size / dt / 1e6:.0f} MB/s of FASTQ ({size / 1e6:.0f} MB in, "
          f"{asz / 1e6:.0f} MB out)  [{line}]", flush=True)
    dt, line = run("-d", "-o", os.path.join(d, "back_" + cont), out + "_1.scalcen", "--patterns-bin", pbin)
    print(f"decompress ({cont}): {dt:.2f} s wall = {size / dt / 1e6:.0f} MB/s of FASTQ  [{line}]", flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "nodigest":  # (the record digests are a Python loop: minutes at 50 M records)
    sys.exit(0)
want = digest(fq)
for cont in ("no", "gz"):
    got = digest(os.path.join(d, f"back_{cont}_1.fastq"))
    print(f"round trip ({cont}): {got[0]} records, multiset {'equal' if got == want else 'DIFFERENT'}", flush=True)
    assert got == want
