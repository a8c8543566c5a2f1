#!/usr/bin/env python3
"""BASELINE.json configs[4]'s workload on ONE GPU: N x 150 bp single-end with the lossy quality map (-p 30) through the `scalce`
binary (streaming host), decompressed again, and the decompressed records compared -- as a multiset, tools/fastq_digest.c --
with what a lossy archive of this input MUST decode to: every record of the input through q' = map[q] - offset, 'N' -> 0, a
base whose q' is 0 -> 'N', quality character q' + offset (qualities.cpp:99-183, decompress.cpp:340-355).  The map is
quality_mapping_init's own (qualities.cpp:99-174 through scalce_qmap_init) on the first 100 000 records, as the binary takes it.
usage: tools/full_size_lossy.py [READS=200000000] [DIR=/dev/shm/scalce_c5] [percentage=30]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
BASE = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm/scalce_c5"
PCT = sys.argv[3] if len(sys.argv) > 3 else "30"
L = int(os.environ.get("L", "150"))
os.makedirs(BASE, exist_ok=True)
D = tempfile.mkdtemp(prefix="scalce_c5_", dir=BASE)
tmp = tempfile.mkdtemp()


def main():
    from scalce_amd import format as fmt
    dig = os.path.join(tmp, "fastq_digest")
    subprocess.run(["gcc", "-O2", "-msse4.2", "-o", dig, os.path.join(R, "tools", "fastq_digest.c")], check=True)
    cli = os.path.join(R, "scalce_amd", "bin", "scalce")
    pbin = os.path.join(R, "tests", "golden", "patterns.bin")
    f1 = os.path.join(D, "in_1.fq")
    t0 = time.time()
    subprocess.run([sys.executable, os.path.join(R, "tools", "gen_fastq.py"), str(N), str(L), f1, "51"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    size = os.path.getsize(f1)
    print(f"generated {N} x {L} bp in {time.time() - t0:.0f} s: {size / 1e9:.1f} GB of FASTQ", flush=True)
    with open(f1, "rb") as f:
        head = f.read(100000 * (2 * L + 40))
    off, vals, Ls = fmt.sample_qmap(head, lossy=int(PCT))
    assert Ls == L
    mapf = os.path.join(tmp, "map.txt")
    open(mapf, "w").write(str(off) + " " + " ".join(str(int(v)) for v in vals) + "\n")
    print("quality map (-p %s): offset %d, %d distinct values for characters 33..126" % (PCT, off, len(set(int(v) for v in vals[33:127]))), flush=True)
    t1 = time.time()
    r = subprocess.run([cli, "-p", PCT, "-c", "no", "-o", os.path.join(D, "arc"), f1, "--patterns-bin", pbin], capture_output=True, text=True)
    dt = time.time() - t1
    if r.returncode:
        print(r.stderr[-2000:])
        return 1
    for line in r.stderr.splitlines():
        if re.search(r"reads found|Time elapsed|Spill|Original size", line):
            print("   ", line.strip())
    asz = sum(os.path.getsize(os.path.join(D, f"arc_1.scalce{e}")) for e in "nrq")
    print(f"compress: {dt:.1f} s wall = {size / dt / 1e6:.0f} MB/s of FASTQ, archive {asz / 1e9:.2f} GB", flush=True)
    want = subprocess.run([dig, "--lossy", mapf, f1], capture_output=True, text=True).stdout.strip()
    plain = subprocess.run([dig, f1], capture_output=True, text=True).stdout.strip()
    os.remove(f1)
    t2 = time.time()
    r = subprocess.run([cli, "-d", "-o", os.path.join(D, "back"), os.path.join(D, "arc_1.scalcen"), "--patterns-bin", pbin], capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-2000:])
        return 1
    dt2 = time.time() - t2
    for line in r.stderr.splitlines():
        if "Time elapsed" in line or "Process:" in line:
            print("   " + line, flush=True)
    print(f"decompress: {dt2:.1f} s wall = {size / dt2 / 1e6:.0f} MB/s of FASTQ", flush=True)
    got = subprocess.run([dig, os.path.join(D, "back_1.fastq")], capture_output=True, text=True).stdout.strip()
    print("input as it is             :", plain)
    print("input through the lossy map:", want)
    print("decompressed               :", got)
    ok = want == got and want != "" and plain != want
    print("LOSSY ROUND TRIP:", "the archive decodes to exactly the input's records through the quality map" if ok else "DIFFERENT")
    return 0 if ok else 1


if __name__ == "__main__":
    try:
        rc = main()
    finally:
        shutil.rmtree(D, ignore_errors=True)
        shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(rc)
