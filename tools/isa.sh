#!/bin/bash
# usage: tools/isa.sh <kernel-name-substring>   -> /tmp/isa_<name>.s (device ISA of one kernel)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R/scalce_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S -o /tmp/scalce_all.s scalce_hip.hip --cuda-device-only -w
sym=$(grep -E "^_Z.*$1.*:" /tmp/scalce_all.s | head -1 | cut -d: -f1)
awk -v s="$sym:" '$1==s,/s_endpgm/' /tmp/scalce_all.s > /tmp/isa_$1.s
grep -A30 "\.name:.*$sym" /tmp/scalce_all.s | grep -E "vgpr_count|sgpr_count|lds_size|scratch" | head -5 || true
wc -l /tmp/isa_$1.s
