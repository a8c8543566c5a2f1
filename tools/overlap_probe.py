#!/usr/bin/env python3
"""Diagnostic: time each front stage of shard B alone and while shard A's arithmetic coder runs on another stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from scalce_amd import host, synth_gpu, format as fmt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
L = 100
dev = torch.device("cuda", 0)
blob = open(os.path.join(ROOT, "tests", "golden", "patterns.bin"), "rb").read()
ctx = host.Context(0, patterns_bin=blob)
text = synth_gpu.fastq_on_device(n, L, dev, seed=20261003, first_index=0)
nbytes = text.numel()
off, vals, Ls = fmt.sample_qmap(text[: 100000 * (2 * L + 20)].cpu().numpy().tobytes())
A = host.Batch(ctx, L, max_reads=n + 8, max_text=nbytes + 64, qmap=[(off, vals), (off, vals)])
B = host.Batch(ctx, L, max_reads=n + 8, max_text=nbytes + 64, qmap=[(off, vals), (off, vals)])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for b in (A, B):
    b.compress(text.data_ptr(), nbytes, None, 0, s1.cuda_stream); b.finish(s1.cuda_stream)

def stages(b, s):
    out = {}
    def t(name, f):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); f(); e1.record(s); e1.synchronize(); out[name] = round(e0.elapsed_time(e1), 2)
    t("ingest", lambda: b.ingest(0, text.data_ptr(), nbytes, s.cuda_stream))
    t("quality", lambda: b.quality(s.cuda_stream))
    t("tokenize", lambda: b.tokenize(None, s.cuda_stream))
    t("order", lambda: b.order(s.cuda_stream))
    t("emit", lambda: b.emit(s.cuda_stream))
    return out

print("alone      :", stages(B, s2))
A.front(text.data_ptr(), nbytes, None, 0, s1.cuda_stream)
GROUP = int(os.environ.get("PROBE_GROUP", "1"))
if GROUP > 1:
    A2 = host.Batch(ctx, L, max_reads=n + 8, max_text=nbytes + 64, qmap=[(off, vals), (off, vals)])
    A2.compress(text.data_ptr(), nbytes, None, 0, s1.cuda_stream); A2.finish(s1.cuda_stream)
    A2.front(text.data_ptr(), nbytes, None, 0, s1.cuda_stream)
t0 = time.perf_counter()
if GROUP > 1:
    host.entropy_begin_group([A, A2], s1.cuda_stream, s1.cuda_stream)
else:
    A.entropy_begin(None, s1.cuda_stream)
time.sleep(0.02)
print("beside AC  :", stages(B, s2), "(front took %.0f ms of host time)" % ((time.perf_counter() - t0) * 1e3))
if GROUP > 1:
    print("beside AC 2:", stages(B, s2))
A.finish(s1.cuda_stream)
print("AC total   : %.0f ms" % ((time.perf_counter() - t0) * 1e3))
