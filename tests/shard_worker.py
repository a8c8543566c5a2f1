"""One rank of a sharded run (tests/test_gpu_sharded_cpp.py starts `world` of these as processes sharing the GPU).
argv: JSON {rank, world, shm (name) | rccl (hex id), L, paired, B, ptxt (path or ""), texts [path per mate], qmap (path or ""), out}"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    a = json.loads(sys.argv[1])
    import torch
    from scalce_amd import host
    rank, world, L = a["rank"], a["world"], a["L"]
    if a["ptxt"]:
        ctx = host.Context(0, patterns_text=open(a["ptxt"], "rb").read())
    else:
        ctx = host.Context(0, patterns_bin=open(os.path.join(ROOT, "tests", "golden", "patterns.bin"), "rb").read())
    if a.get("rccl"):
        comm = host.Comm(0, world, rank, unique_id=bytes.fromhex(a["rccl"]))
    else:
        comm = host.Comm(0, world, rank, shm_name=a["shm"])
    texts = [open(p, "rb").read() for p in a["texts"]]
    dev = [torch.frombuffer(bytearray(t) if len(t) else bytearray(16), dtype=torch.uint8).to("cuda:0") for t in texts]
    qm = None
    if a["qmap"]:
        q = np.load(a["qmap"])
        qm = [(int(q["off"][m]), q["vals"][m]) for m in range(2)]
    paired = a["paired"]
    b = host.Batch(ctx, L, max_reads=max(1024, len(texts[0]) // (2 * L + 7) + 8), max_text=max(len(t) for t in texts) + 64, paired=paired,
                   read_len2=L, qmap=qm, bucket_set_size=a["B"])
    res = host.sharded_compress(comm, ctx, b, dev[0].data_ptr(), len(texts[0]), dev[1].data_ptr() if paired else None,
                                len(texts[1]) if paired else 0)
    nb1 = res.nb1
    out = dict(counts=np.ctypeslib.as_array(res.counts, shape=(world * nb1,)).copy().reshape(world, nb1),
               name_bytes=np.ctypeslib.as_array(res.name_bytes, shape=(world * nb1,)).copy().reshape(world, nb1),
               reads=b.output(host.OUT_READS, 0), names=b.output(host.OUT_NAMES, 0), qual=b.output(host.OUT_QUAL, 0),
               table=b.output(host.OUT_TABLE, 0, np.uint32), tokens=b.output(host.OUT_TOKENS, 0, np.int32),
               meta=np.array([res.reads_total, res.first_read, res.reads_local, res.rounds, res.sweeps, res.chunks_total,
                              res.moved_in[0], res.moved_in[1]], dtype=np.int64))
    if paired:
        out.update(reads2=b.output(host.OUT_READS, 1), qual2=b.output(host.OUT_QUAL, 1), table2=b.output(host.OUT_TABLE, 1, np.uint32))
    np.savez(a["out"], **out)
    host.shard_result_free(res)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
