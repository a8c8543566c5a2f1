#!/bin/bash
# round 5, full-size evidence: (1) 30 M x 150 bp single-end -p 30 against the reference's own compress(); (2) BASELINE configs[2],
# 200 M pairs x 150 bp through the binary and back; (3) configs[4]'s workload on one GPU: 200 M x 150 bp -p 30, compress + decompress
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5_evidence
python tools/full_size_ref_check.py --se 0 --pe 0 --lossy 30000000 --log gpurun_out/r5_evidence/full_size_lossy150.log --json gpurun_out/r5_evidence/full_size_lossy150.json 2>&1 | tail -12
echo "== 200 M pairs"
timeout -k 10 900 python tools/full_size_paired.py 200000000 > gpurun_out/r5_evidence/c3_full_size_200m_pairs.log 2>&1; echo "rc $?"; tail -12 gpurun_out/r5_evidence/c3_full_size_200m_pairs.log
