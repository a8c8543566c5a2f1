#!/bin/bash
# timing experiment: which part of emit_reads_k<true> costs what (SCALCE_EMIT_ABLATE drops parts; outputs are wrong)
cd "$(dirname "$0")/.."
for v in "" q c p qc qcp; do
  SCALCE_EMIT_ABLATE=$v bash tools/kernel_times.sh abl_$v $( [ -n "$v" ] && echo SCALCE_EMIT_ABLATE=$v ) >/dev/null 2>&1
  echo "ablate '$v': $(grep -E 'emit_reads_k|trigram_pass_k|ingest_tiles2' gpurun_out/abl_${v}_alone_kernel_stats.csv | cut -d, -f1,4 | tr '\n' ' ')"
done
