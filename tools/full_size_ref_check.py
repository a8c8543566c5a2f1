#!/usr/bin/env python3
"""Byte-identity with the REFERENCE'S OWN compress() at full size.

The fixtures under tests/golden pin the product to the reference at 30 000 reads; the bench's sample at 1.5 M.  Neither
reaches what BASELINE configs[1] is made of: three spill chunks (-B 4G, compress.cpp:708-715), a shrink factor of 2
(compress.cpp:297-313), more than 2^32 quality symbols, nine million tie reads (reads.cpp:246,420-421).  This tool runs
both sides on the SAME file at that size and compares SHA-256 of every archive file:

  se   : N x 100 bp single-end (default 50 M = configs[1]; the text of bench.py's shard, seed 20261003)
  pe   : P pairs x 150 bp, -r (default 30 M pairs: >= 3 chunks, factor 2)
  lossy: M x 150 bp single-end with -p 30 (round 5; default off, 30 M: >= 3 chunks, factor 2 -- the lossy quality map of
         qualities.cpp:117-174 and the table scaling of compress.cpp:297-313 at a size where both matter; BASELINE configs[4]'s
         workload on one GPU)

  reference : oracle/_ref/ref_full compress -c no -T 1   (the reference's own objects; ~42 MB/s)
  product   : scalce_amd/bin/scalce -c no

usage: tools/full_size_ref_check.py [--se N] [--pe P] [--lossy M] [--dir /dev/shm/scalce_refcheck] [--log profiles/r04_full_size_ref_check.log]
Both reference runs go side by side (one thread each); a progress line a minute keeps the GPU box's watchdog quiet.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "ref_full")
CLI = os.path.join(ROOT, "scalce_amd", "bin", "scalce")
PBIN = os.path.join(ROOT, "tests", "golden", "patterns.bin")


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(64 << 20)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def gen(n, L, path, seed, first=0, pair_suffix=None):
    """bench.py's generator, chunk by chunk (a chunk's seed = seed + first record, as tools/gen_fastq.py)."""
    import torch

    from scalce_amd import synth_gpu
    dev = torch.device("cuda", 0)
    if pair_suffix is None and n <= 60_000_000:
        t = synth_gpu.fastq_on_device(n, L, dev, seed=seed, first_index=first)  # exactly bench.py's shard
        with open(path, "wb") as f:
            step = 1 << 30
            for a in range(0, t.numel(), step):
                f.write(memoryview(t[a:a + step].cpu().numpy()))
        del t
        torch.cuda.empty_cache()
        return os.path.getsize(path)
    step = 8_000_000
    with open(path, "wb") as f:
        for a in range(0, n, step):
            m = min(step, n - a)
            t = synth_gpu.fastq_on_device(m, L, dev, seed=seed + a, first_index=a)
            f.write(memoryview(t.cpu().numpy()))
            del t
    torch.cuda.empty_cache()
    return os.path.getsize(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--se", type=int, default=50_000_000)
    ap.add_argument("--pe", type=int, default=30_000_000)
    ap.add_argument("--lossy", type=int, default=0, help="reads of the 150 bp single-end -p 30 case (0 = skip)")
    ap.add_argument("--dir", default="/dev/shm/scalce_refcheck")
    ap.add_argument("--log", default=None)
    ap.add_argument("--json", default=None, help="write the result object here as well")
    args = ap.parse_args()
    assert os.path.exists(REF), "oracle/_ref/ref_full is not built (make -C oracle, needs /root/reference)"
    assert os.path.exists(CLI), "scalce binary not built"
    os.makedirs(args.dir, exist_ok=True)
    d = tempfile.mkdtemp(prefix="run_", dir=args.dir)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    result = {}
    try:
        cases = []
        if args.se > 0:
            f = os.path.join(d, "se_1.fq")
            t0 = time.time()
            sz = gen(args.se, 100, f, 20261003)
            say(f"se: generated {args.se} x 100 bp, {sz} bytes, {time.time() - t0:.0f} s")
            cases.append(("se", [f], [], sz, 1, args.se, 100))
        if args.pe > 0:
            f1, f2 = os.path.join(d, "pe_1.fq"), os.path.join(d, "pe_2.fq")
            t0 = time.time()
            sz = gen(args.pe, 150, f1, 41, pair_suffix=1) + gen(args.pe, 150, f2, 42, pair_suffix=2)
            say(f"pe: generated {args.pe} pairs x 150 bp, {sz} bytes, {time.time() - t0:.0f} s")
            cases.append(("pe", [f1], ["-r"], sz, 2, args.pe, 150))
        if args.lossy > 0:
            f = os.path.join(d, "lossy_1.fq")
            t0 = time.time()
            sz = gen(args.lossy, 150, f, 51, pair_suffix=0)
            say(f"lossy: generated {args.lossy} x 150 bp, {sz} bytes, {time.time() - t0:.0f} s")
            cases.append(("lossy", [f], ["-p", "30"], sz, 1, args.lossy, 150))
        # the reference, both cases side by side
        procs = []
        for name, files, flags, sz, mates, n, L in cases:
            tmpd = os.path.join(d, "tmp_" + name)
            cmd = [REF, "compress", PBIN, files[0], os.path.join(d, "ref_" + name), "-c", "no", "-T", "1", "-t", tmpd, *flags]
            err = open(os.path.join(d, f"ref_{name}.err"), "wb")
            procs.append((name, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=err), time.time(), err))
        # the product meanwhile (seconds)
        for name, files, flags, sz, mates, n, L in cases:
            t0 = time.time()
            r = subprocess.run([CLI, "-c", "no", *flags, "-o", os.path.join(d, "hip_" + name), files[0], "--patterns-bin", PBIN],
                               capture_output=True, text=True)
            dt = time.time() - t0
            if r.returncode:
                say(f"{name}: scalce FAILED: {r.stderr[-1500:]}")
                return 1
            info = [ln.strip() for ln in r.stderr.splitlines() if any(k in ln for k in ("reads found", "Spill", "shrink", "chunks"))]
            say(f"{name}: scalce -c no {dt:.1f} s = {sz / dt / 1e6:.0f} MB/s (beside the reference runs)  {info}")
        done = {}
        while len(done) < len(procs):
            time.sleep(20)
            for name, p, t0, err in procs:
                if name not in done and p.poll() is not None:
                    done[name] = (p.returncode, time.time() - t0)
                    err.close()
            say("  ... reference running: " + ", ".join(f"{name} {'done' if name in done else '%.0f s' % (time.time() - t0)}" for name, p, t0, err in procs))
        ok_all = True
        for name, files, flags, sz, mates, n, L in cases:
            rc, dt = done[name]
            errtxt = open(os.path.join(d, f"ref_{name}.err"), "rb").read().decode(errors="replace")
            if rc:
                say(f"{name}: reference FAILED rc {rc}: {errtxt[-1500:]}")
                return 1
            info = [ln.strip() for ln in errtxt.splitlines() if any(k in ln for k in ("shrink factor", "Temp file", "reads found", "emp"))]
            say(f"{name}: reference compress() -T 1: {dt:.0f} s = {sz / dt / 1e6:.1f} MB/s  {info[:8]}")
            res = {"reads" if mates == 1 else "pairs": n, "length": L, "input_bytes": sz, "reference_seconds": round(dt, 1),
                   "factor": 1 + (n * L) // ((1 << 32) - 1), "files": {}}
            for m in range(1, mates + 1):
                for e in "nrq":
                    a, b = os.path.join(d, f"hip_{name}_{m}.scalce{e}"), os.path.join(d, f"ref_{name}_{m}.scalce{e}")
                    ha, hb = sha(a), sha(b)
                    same = ha == hb
                    ok_all &= same
                    res["files"][f"{m}.scalce{e}"] = {"same": same, "bytes": os.path.getsize(b), "sha256": hb}
                    say(f"{name}: _{m}.scalce{e}  {os.path.getsize(a):>12} / {os.path.getsize(b):>12} bytes  {'IDENTICAL' if same else 'DIFFERENT'}  {hb[:16]}")
            res["ok"] = all(v["same"] for v in res["files"].values())
            result[name] = res
        say("FULL-SIZE REFERENCE CHECK: " + ("every archive file byte-identical with the reference's own compress()" if ok_all else "DIFFERENT"))
        result["ok"] = ok_all
        return 0 if ok_all else 1
    finally:
        shutil.rmtree(d, ignore_errors=True)
        if args.log:
            os.makedirs(os.path.dirname(os.path.abspath(args.log)), exist_ok=True)
            open(args.log, "w").write("\n".join(lines) + "\n")
        if args.json:
            json.dump(result, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
