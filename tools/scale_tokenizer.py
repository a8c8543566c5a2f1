#!/usr/bin/env python3
"""Diagnostic: the tokenizer against a core table of realistic size (1 M cores of 12-32 bases: tests/bigtable.py) -- parity
with the oracle's trie walk and the time of the tokenize stage.  usage: tools/scale_tokenizer.py [reads=5000000] [L=100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bigtable
import oraclelib as O
from scalce_amd import host, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.perf_counter()
blob, vals = bigtable.build()
print(f"table: {sum(len(v) for _, v in vals)} cores, {len(blob)} bytes, built in {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
ctx = host.Context(0, patterns_bin=blob)
print(f"device automaton: {ctx.n_states} states, {time.perf_counter() - t0:.1f} s", flush=True)
bases = bigtable.reads_with_cores(n, L, vals)
quals = np.full((n, L), ord("I"), dtype=np.uint8)
fq = synth.fastq_bytes_fast(bases, quals)
t = torch.frombuffer(bytearray(fq), dtype=torch.uint8).to("cuda:0")
b = host.Batch(ctx, L, n + 8, len(fq) + 64)
for rep in range(2):
    b.stage_reset(True)
    b.compress(t.data_ptr(), len(fq)); b.finish()
print("stage ms:", {s: round(v[0], 2) for s, v in b.stage_ms().items()}, b.stats(), flush=True)
t0 = time.perf_counter()
trie = O.Trie(blob=blob)
pat, end = trie.tokenize(bases)
print(f"oracle trie + tokenize: {time.perf_counter() - t0:.1f} s; bucketed {100.0 * (pat >= 0).mean():.1f} %", flush=True)
tok = b.output(host.OUT_TOKENS, 0, np.int32).reshape(-1, 2)
ok = bool((tok[:, 0] == pat).all() and (tok[:, 1] == end).all())
print("tokens equal the oracle's:", ok)
perm = trie.order(bases, pat, end)
print("order equal the oracle's:", bool((b.output(host.OUT_PERM, 0, np.uint32) == perm).all()))
sys.exit(0 if ok else 1)
