import os, sys, struct
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oraclelib as O
from scalce_amd import host, synth
from gpu_util import device_bytes
blob = open("/root/repo/tests/golden/patterns.bin", "rb").read()
ctx = host.Context(0, patterns_bin=blob)
rng = np.random.default_rng(5)
n, L = 120_000, 100
bases, quals = synth.reads_and_quals(n, L, seed=32)
quals = (rng.integers(0, 80, size=(n, L)) + 33).astype(np.uint8)
fq = synth.fastq_bytes_fast(bases, quals)
t = device_bytes(fq)
res = {}
for v in ("1", "1n", "4"):
    os.environ["SCALCE_AC_BLOCKS_PER_WG"] = v[0]
    if v.endswith("n"): os.environ["SCALCE_AC_NO_ELECTION"] = "1"
    else: os.environ.pop("SCALCE_AC_NO_ELECTION", None)
    b = host.Batch(ctx, L, n + 8, len(fq) + 64)
    b.compress(t.data_ptr(), len(fq)); b.finish()
    res[v] = b.output(host.OUT_QUAL, 0).copy()
    qs = b.output(host.OUT_QSTREAM, 0).copy(); table = b.output(host.OUT_TABLE, 0, np.uint32).copy()
want = O.AcStat(table).encode_stream(qs)
for v in ("1", "1n", "4"):
    g = res[v]
    neq = np.flatnonzero(g[:min(len(g), len(want))] != want[:min(len(g), len(want))])
    sz0 = struct.unpack_from("<I", want, 0)[0]
    print("variant", v, "len", len(g), "want", len(want), "first diffs", neq[:5], "block0 size", sz0, "nblocks", (n*L + 10485759)//10485760)
