#!/usr/bin/env python3
"""bench.py's table_scale leg on its own (1 M cores of 12-32 bases), for several numbers of reads: what of the tokenize stage's
time is per read and what is per table.  usage: tools/table_scale_probe.py [reads ...]   (under rocprofv3 --kernel-trace --stats
for the kernels behind it)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

for n in [int(x) for x in sys.argv[1:]] or [2_000_000, 8_000_000]:
    r = bench.table_scale_leg(torch.device("cuda", 0), reads=n)
    print(json.dumps({k: r[k] for k in ("reads", "tokenize_stage_ms", "ns_per_read", "tie_reads")}), flush=True)
