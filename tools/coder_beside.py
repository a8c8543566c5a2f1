#!/usr/bin/env python3
"""Diagnostic: how long does one coder launch (three shards, eight blocks per chain wave) take while ONE kind of front
stage runs beside it over and over?  Tells which front kernels slow the coder's helper waves down."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scalce_amd import host, synth_gpu, format as fmt

n, L = 50_000_000, 100
dev = torch.device("cuda", 0)
ctx = host.Context(0, patterns_bin=open(os.path.join(ROOT, "tests", "golden", "patterns.bin"), "rb").read())
text = synth_gpu.fastq_on_device(n, L, dev, seed=20261003, first_index=0)
nbytes = text.numel()
off, vals, Ls = fmt.sample_qmap(text[: 100000 * (2 * L + 20)].cpu().numpy().tobytes())
mk = lambda: host.Batch(ctx, L, max_reads=n + 8, max_text=nbytes + 64, qmap=[(off, vals), (off, vals)])
grp = [mk() for _ in range(3)]
B = mk()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for b in grp + [B]:
    b.compress(text.data_ptr(), nbytes, None, 0, s1.cuda_stream); b.finish(s1.cuda_stream)
stages = {
    "nothing": None,
    "ingest": lambda: B.ingest(0, text.data_ptr(), nbytes, s2.cuda_stream),
    "quality": lambda: B.quality(s2.cuda_stream),
    "tokenize": lambda: B.tokenize(None, s2.cuda_stream),
    "tok begin": lambda: B.tokenize_begin(s2.cuda_stream),   # DFA walk, tie candidates, events sorted by bucket
    "tok sweeps": lambda: [B.tokenize_sweep(None, s2.cuda_stream) for _ in range(10)],  # ten tie-break sweeps

    "order": lambda: B.order(s2.cuda_stream),
    "emit": lambda: B.emit(s2.cuda_stream),
}
only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
for name, f in stages.items():
    if only and name not in only:
        continue
    for b in grp:
        b.front(text.data_ptr(), nbytes, None, 0, s1.cuda_stream)
    torch.cuda.synchronize()
    host.entropy_begin_group(grp, s1.cuda_stream, s1.cuda_stream)
    ev = torch.cuda.Event(); ev.record(s1)
    t0 = time.perf_counter(); reps = 0
    while not ev.query():
        if f is None:
            time.sleep(0.005)
        else:
            f(); s2.synchronize(); reps += 1
    dt = (time.perf_counter() - t0) * 1e3
    for b in grp:
        b.finish(s2.cuda_stream)
    print(f"coder beside {name:10s}: {dt:6.0f} ms  ({reps} runs of the stage meanwhile)", flush=True)
