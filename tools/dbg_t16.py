import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import oraclelib as O
from scalce_amd import host, synth
from gpu_util import hip_compress, oracle_streams, device_bytes
blob = open("/root/repo/tests/golden/patterns.bin","rb").read()
ctx = host.Context(0, patterns_bin=blob)
trie = O.Trie(blob=blob)
rng = np.random.default_rng(5)
# precondition: what test_grouped_coder_launch_equals_separate_launches does
specs = [(230_000, 100, 31), (120_000, 100, 32), (70_000, 36, 33)]
texts = []
for n, L, seed in specs:
    bases, quals = synth.reads_and_quals(n, L, seed=seed)
    if seed == 32:
        quals = (rng.integers(0, 80, size=(n, L)) + 33).astype(np.uint8)
    fq = synth.fastq_bytes_fast(bases, quals)
    texts.append((device_bytes(fq), len(fq), n, L))
pre = sys.argv[2] if len(sys.argv) > 2 else "group"
if pre in ("alone", "both"):
    for variant in ("1", "4"):
        os.environ["SCALCE_AC_BLOCKS_PER_WG"] = variant
        for t, nb, n, L in texts:
            b = host.Batch(ctx, L, n + 8, nb + 64); b.compress(t.data_ptr(), nb); b.finish()
    del os.environ["SCALCE_AC_BLOCKS_PER_WG"]
if pre in ("group", "both"):
    group = []
    for t, nb, n, L in texts:
        b = host.Batch(ctx, L, n + 8, nb + 64); b.front(t.data_ptr(), nb); group.append(b)
    host.entropy_begin_group(group)
    for b in group: b.finish()
    del group, b
for bpw in ("8",):
    os.environ["SCALCE_AC_BLOCKS_PER_WG"] = bpw
    os.environ["SCALCE_AC_TEST_POISON"] = sys.argv[1] if len(sys.argv) > 1 else "3"
    bases, quals = synth.reads_and_quals(150_000, 100, seed=77)
    fq = synth.fastq_bytes_fast(bases, quals)
    b = hip_compress(ctx, fq, 100)
    ref = oracle_streams(trie, bases, quals, 33, None)
    table = b.output(host.OUT_TABLE, 0, np.uint32)
    want_qs = ref["qp"][ref["perm"]].reshape(-1)
    enc = b.output(host.OUT_QUAL, 0)
    want = O.AcStat(table).encode_stream(want_qs)
    print("len", len(enc), len(want))
    neq = np.flatnonzero(enc != want)
    print("ndiff", len(neq), "first", neq[:8], "last", neq[-3:] if len(neq) else None)
    if len(neq):
        import struct
        print("enc ", enc[:48].tobytes().hex())
        print("want", want[:48].tobytes().hex())
        s0e = struct.unpack("<I", enc[:4].tobytes())[0]
        print("enc block0 size", s0e, "st", b.stats())
        # would the oracle produce these bytes from the INPUT-order stream?
        alt = O.AcStat(table).encode_stream(ref["qp"].reshape(-1))
        print("input-order oracle equals enc:", len(alt) == len(enc) and bool((alt == enc).all()))
        qs = b.output(host.OUT_QSTREAM, 0)
        print("qstream equals reordered:", bool((qs == want_qs).all()))
        s0 = struct.unpack("<I", want[:4].tobytes())[0]
        print("block0 size", s0, "diffs in block0", int((neq < 4 + s0).sum()), "in block1", int((neq >= 4 + s0).sum()))
        d = neq[neq < 4 + s0] - 4
        print("word index of first diffs", (d[:10] // 4), "hist of diff mod 256 words:", np.bincount((d // 4) % 64, minlength=64)[:16])
