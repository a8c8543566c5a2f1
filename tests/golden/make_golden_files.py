#!/usr/bin/env python3
"""Whole-file golden hashes from the REAL reference pipeline (oracle/_ref/ref_full).

ref_full is the reference's own compress() / decompress() (every source but main.cpp, compiled where
it lies: see oracle/Makefile).  For each case below this script regenerates the input from its seed
(tests/filecases.py), runs the reference at -T 1, and records the SHA-256 of every .scalce{n,r,q}
file it wrote (content hash after gunzip for -c gz) and of the FASTQ its decompress() restores from
them.  Run in the build container only (needs /root/reference); writes tests/golden/files.json.
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import filecases as F  # noqa: E402

if __name__ == "__main__":
    if not os.path.exists(F.REF_FULL):
        sys.exit("oracle/_ref/ref_full missing: run `make -C oracle` where /root/reference exists")
    out = {}
    for name in F.CASES:
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        with tempfile.TemporaryDirectory() as d:
            F.write_inputs(name, d)
            F.run_tool("ref", name, d, "ref")
            out[name] = F.hash_outputs(name, d, "ref")
            print(name, {k: v[:12] for k, v in out[name].items()})
    path = os.path.join(HERE, "files.json")
    if len(sys.argv) > 1 and os.path.exists(path):
        old = json.load(open(path))
        old.update(out)
        out = old
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
