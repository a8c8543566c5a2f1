"""-m gpu: scalce_sharded_compress (C++ host, comm.cpp / sharded.cpp).  `world` processes share the GPU and talk through
the shared-memory rehearsal transport; the pieces they produce, assembled bucket by bucket in rank order, are byte for byte
the archive ONE batch makes of the whole input with the same -B -- tokens incl. the run-wide tie-break, spill chunks cut on
run-wide record sizes (rank boundaries move to chunk boundaries: records change owner), names, quality table, coder blocks
cut on the run-wide stream.  One case runs the RCCL transport itself (world 1: every collective is a real RCCL call)."""
import json
import os
import struct
import subprocess
import sys
import uuid

import numpy as np
import pytest

import oraclelib as O
from scalce_amd import format as fmt
from scalce_amd import host, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def assemble(ctx, outs, L, paired):
    """Pieces of the ranks -> payloads of one archive (what a writer does with pwrite at offsets)."""
    world = len(outs)
    C = outs[0]["counts"].astype(np.int64)
    NB = outs[0]["name_bytes"].astype(np.int64)
    order = ctx.bucket_patterns()
    levels = np.array([0 if p == host.ROOT_CORE else len(ctx.pattern(int(p))) for p in order], dtype=np.int64)
    recsz = (L - levels + 3) // 4 + (2 if L > 255 else 1)
    reads, names = [], []
    off = [np.cumsum(np.where(C[r] > 0, 12 + C[r] * recsz, 0)) - np.where(C[r] > 0, 12 + C[r] * recsz, 0) for r in range(world)]
    noff = [np.cumsum(NB[r]) - NB[r] for r in range(world)]
    Cg = C.sum(axis=0)
    for b in np.flatnonzero(Cg):
        reads.append(struct.pack("<iq", int(order[b]), int(Cg[b])))
        for r in range(world):
            if C[r][b]:
                a = int(off[r][b]) + 12
                reads.append(outs[r]["reads"][a:a + int(C[r][b] * recsz[b])].tobytes())
                names.append(outs[r]["names"][int(noff[r][b]):int(noff[r][b] + NB[r][b])].tobytes())
    res = dict(reads=b"".join(reads), names=b"".join(names), qual=b"".join(o["qual"].tobytes() for o in outs))
    if paired:
        res["qual2"] = b"".join(o["qual2"].tobytes() for o in outs)
        # mate 2: bare records in mate 1's order -> the same interleave, 38-byte rows
        w = (L + 3) // 4
        r2 = []
        first = [np.cumsum(C[r]) - C[r] for r in range(world)]
        for b in np.flatnonzero(Cg):
            for r in range(world):
                if C[r][b]:
                    a = int(first[r][b]) * w
                    r2.append(outs[r]["reads2"][a:a + int(C[r][b]) * w].tobytes())
        res["reads2"] = b"".join(r2)
    return res


def run_ranks(tmp_path, world, texts_per_rank, L, paired, B, ptxt=None, qmap=None, rccl=False):
    shm = "/scalce_test_" + uuid.uuid4().hex[:12]
    rid = host.Comm.unique_id().hex() if rccl else ""
    procs = []
    for r in range(world):
        paths = []
        for m, t in enumerate(texts_per_rank[r]):
            p = tmp_path / f"rank{r}_{m + 1}.fq"
            open(p, "wb").write(t)
            paths.append(str(p))
        args = dict(rank=r, world=world, shm=shm, rccl=rid, L=L, paired=paired, B=B, ptxt=str(ptxt) if ptxt else "", texts=paths,
                    qmap=str(qmap) if qmap else "", out=str(tmp_path / f"out{r}.npz"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "shard_worker.py"), json.dumps(args)],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{logs[r][-3000:]}"
    return [dict(np.load(tmp_path / f"out{r}.npz")) for r in range(world)]


def split_records(text, cuts):
    """text of n records -> pieces at record indices `cuts`."""
    nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
    ends = [0] + [int(nl[4 * c - 1]) + 1 if c else 0 for c in cuts] + [len(text)]
    return [text[ends[i]:ends[i + 1]] for i in range(len(ends) - 1)]


@pytest.mark.parametrize("case", ["se100_w2", "pe150_lossy_w3", "ties36_w4", "tiny_rank_w3", "empty_rank_w3", "rccl_w1", "uncut_w3", "uncut_pe_w2"])
def test_sharded_archive_equals_one_batch(case, tmp_path, patterns_blob):
    from gpu_util import device_bytes
    paired, L, n, world, B, ptxt, lossy, rccl = False, 100, 60000, 2, 1_500_000, None, 0, False
    if case == "pe150_lossy_w3":
        paired, L, n, world, B, lossy = True, 150, 30000, 3, 2_000_000, 30
    elif case == "ties36_w4":
        import itertools
        L, n, world, B = 36, 40000, 4, 500_000
        ptxt = tmp_path / "p.txt"
        open(ptxt, "w").write("\n".join("".join(x) for x in itertools.product("ACGT", repeat=4)) + "\n")
    elif case in ("tiny_rank_w3", "empty_rank_w3"):
        world, n, B = 3, 50000, 1_000_000
    elif case == "rccl_w1":
        world, n, rccl = 1, 30000, True
    elif case == "uncut_w3":      # -B larger than the run: ONE chunk, every bucket merged across the ranks (all rows to rank 0)
        world, n, B = 3, 45000, 1 << 30
    elif case == "uncut_pe_w2":
        paired, L, n, world, B = True, 150, 20000, 2, 1 << 30
    bases, quals = synth.reads_and_quals(n, L, seed=131, dup_frac=0.15, n_frac=0.003)
    texts = [synth.fastq_bytes_fast(bases, quals, prefix="p." if paired else "s.", suffix="/1" if paired else "")]
    if paired:
        b2, q2 = synth.reads_and_quals(n, L, seed=132, n_frac=0.003)
        texts.append(synth.fastq_bytes_fast(b2, q2, prefix="p.", suffix="/2"))
    cuts = [int(n * (r + 1) / world) + d for r, d in zip(range(world - 1), (7, -13, 5))]
    if case == "tiny_rank_w3":
        cuts = [3, 40000]   # rank 0 holds three records: they all move to... wherever the first chunk boundary says
    if case == "empty_rank_w3":
        cuts = [30000, 30000]   # the middle rank starts with no text at all
    pieces = [split_records(t, cuts) for t in texts]
    per_rank = [[pieces[m][r] for m in range(len(texts))] for r in range(world)]
    qm, qpath = None, None
    if lossy:
        qm = [fmt.sample_qmap(t, lossy=lossy)[:2] for t in texts]
        qpath = tmp_path / "qmap.npz"
        np.savez(qpath, off=np.array([q[0] for q in qm]), vals=np.stack([q[1] for q in qm]))
    outs = run_ranks(tmp_path, world, per_rank, L, paired, B, ptxt=ptxt, qmap=qpath, rccl=rccl)
    ctx = host.Context(0, patterns_text=open(ptxt, "rb").read()) if ptxt else host.Context(0, patterns_bin=patterns_blob)
    got = assemble(ctx, outs, L, paired)
    # the same input as ONE batch with the same -B
    dev = [device_bytes(t) for t in texts]
    one = host.Batch(ctx, L, n + 8, max(len(t) for t in texts) + 64, paired=paired, read_len2=L, qmap=qm, bucket_set_size=B)
    one.compress(dev[0].data_ptr(), len(texts[0]), dev[1].data_ptr() if paired else None, len(texts[1]) if paired else 0)
    one.finish()
    meta = np.stack([o["meta"] for o in outs])
    print(case, "chunks", one.stats()["chunks"], "rows per rank", meta[:, 2], "moved in", meta[:, 6:8].tolist(), "rounds", meta[:, 3], "sweeps", meta[:, 4])
    assert meta[0, 0] == n and meta[:, 2].sum() == n and int(meta[0, 5]) == one.stats()["chunks"]
    toks = np.concatenate([o["tokens"].reshape(-1, 2) for o in outs])
    assert (toks == one.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)).all(), "tokens (run-wide tie-break)"
    assert (outs[0]["table"] == one.output(host.OUT_TABLE, 0, np.uint32)).all(), "run-wide quality table"
    assert got["reads"] == one.output(host.OUT_READS, 0).tobytes(), ".scalcer payload"
    assert got["names"] == one.output(host.OUT_NAMES, 0).tobytes(), ".scalcen payload"
    assert got["qual"] == one.output(host.OUT_QUAL, 0).tobytes(), "coder blocks on the run-wide stream"
    if paired:
        assert (outs[0]["table2"] == one.output(host.OUT_TABLE, 1, np.uint32)).all()
        assert got["reads2"] == one.output(host.OUT_READS, 1).tobytes()
        assert got["qual2"] == one.output(host.OUT_QUAL, 1).tobytes()
    if case == "se100_w2":  # and the oracle says the same (the one-batch path is checked against it everywhere else)
        open(tmp_path / "in_1.fq", "wb").write(texts[0])
        O.orc_cli("compress", os.path.join(HERE, "golden", "patterns.bin"), tmp_path / "in_1.fq", tmp_path / "orc", "-B", B)
        assert open(tmp_path / "orc_1.scalcer", "rb").read()[16:] == got["reads"]
        assert open(tmp_path / "orc_1.scalceq", "rb").read()[16 + 2048000 + 8:] == got["qual"]
