#!/bin/bash
# kernel resource table (VGPRs, scratch, spills, LDS, occupancy) of one csrc/*.hip file, as hipcc reports it:
#   tools/kres.sh k_kv.hip [filter]        (CPU only: hipcc cross-compiles gfx950)
cd "$(dirname "$0")/../dint_amd/csrc" || exit 1
F=${1:-k_kv_tatp.hip}; PAT=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -c "$F" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|remark" | sed 's/.*remark: *//; s/ \[-Rpass.*//' \
  | awk '/Function Name/{if(l)print l; l=$0; next}{gsub(/^ +/,""); l=l" | "$0}END{print l}' \
  | sed 's/[a-z_]*\.h*i*p*:[0-9]*:[0-9]*: *//g; s/AGPRs: 0 | //; s/Dynamic Stack: False | //; s/Function Name: //' | grep -E "error|$PAT" | cut -c1-260
