#!/usr/bin/env python3
"""BASELINE.json configs[2] through the product path: N pairs x 150 bp (-r) as two files in /dev/shm, compressed by the
`scalce` binary (streaming host: at full size, 126 GB, the text does not fit HBM), decompressed again, and the decompressed
pairs compared with the input as a multiset (tools/fastq_digest.c: count, sum and xor of a hash per pair).
usage: tools/full_size_paired.py [PAIRS=200000000] [DIR=/dev/shm/scalce_c3] [extra scalce flags ...]"""
import os
import re
import resource
import shutil
import subprocess
import sys
import tempfile
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
D = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm/scalce_c3"
FLAGS = sys.argv[3:]
L = int(os.environ.get("L", "150"))
BASE = D
os.makedirs(BASE, exist_ok=True)
D = tempfile.mkdtemp(prefix="scalce_c3_", dir=BASE)  # a directory of this run's own: only that is removed at the end
tmp = tempfile.mkdtemp()


def main():
    dig = os.path.join(tmp, "fastq_digest")
    subprocess.run(["gcc", "-O2", "-msse4.2", "-o", dig, os.path.join(R, "tools", "fastq_digest.c")], check=True)
    cli = os.path.join(R, "scalce_amd", "bin", "scalce")
    pbin = os.path.join(R, "tests", "golden", "patterns.bin")
    f1, f2 = os.path.join(D, "in_1.fq"), os.path.join(D, "in_2.fq")
    t0 = time.time()
    gens = [subprocess.Popen([sys.executable, os.path.join(R, "tools", "gen_fastq.py"), str(N), str(L), f, str(seed)],
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for f, seed in ((f1, 41), (f2, 42))]
    assert all(g.wait() == 0 for g in gens)
    size = os.path.getsize(f1) + os.path.getsize(f2)
    print(f"generated {N} pairs x {L} bp in {time.time() - t0:.0f} s: {size / 1e9:.1f} GB of FASTQ", flush=True)
    t1 = time.time()
    r = subprocess.run([cli, "-r", "-c", "no", *FLAGS, "-o", os.path.join(D, "arc"), f1, "--patterns-bin", pbin], capture_output=True, text=True)
    dt = time.time() - t1
    if r.returncode:
        print(r.stderr[-2000:])
        return 1
    for line in r.stderr.splitlines():
        if re.search(r"reads found|Time elapsed|Spill|Original size", line):
            print("   ", line.strip())
    asz = sum(os.path.getsize(os.path.join(D, f"arc_{m}.scalce{e}")) for m in (1, 2) for e in "nrq")
    print(f"compress: {dt:.1f} s wall = {size / dt / 1e6:.0f} MB/s of FASTQ, archive {asz / 1e9:.2f} GB, "
          f"host peak RSS of children {resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6:.1f} GB", flush=True)
    want = subprocess.run([dig, f1, f2], capture_output=True, text=True).stdout.strip()  # (not beside the compressor: it would share its memory bandwidth)
    os.remove(f1)
    os.remove(f2)
    t2 = time.time()
    r = subprocess.run([cli, "-d", "-r", "-o", os.path.join(D, "back"), os.path.join(D, "arc_1.scalcen"), "--patterns-bin", pbin],
                       capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-2000:])
        return 1
    dt2 = time.time() - t2
    for line in r.stderr.splitlines():
        if "Time elapsed" in line or "Process:" in line:
            print("   " + line, flush=True)
    print(f"decompress: {dt2:.1f} s wall = {size / dt2 / 1e6:.0f} MB/s of FASTQ", flush=True)
    got = subprocess.run([dig, os.path.join(D, "back_1.fastq"), os.path.join(D, "back_2.fastq")], capture_output=True, text=True).stdout.strip()
    print("input :", want)
    print("output:", got)
    ok = want == got and want != ""
    print("ROUND TRIP:", "pairs multiset-equal" if ok else "DIFFERENT")
    return 0 if ok else 1


if __name__ == "__main__":
    try:
        rc = main()
    finally:  # also on a failure: the 126 GB of input and archive files and the digest binary do not stay behind
        shutil.rmtree(D, ignore_errors=True)
        shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(rc)
