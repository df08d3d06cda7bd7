#!/bin/bash
# tools/isa.sh <file.hip> [extra hipcc flags...]: compile one source of the library exactly as clover_amd/build.py does and leave the gfx950
# assembly in /tmp/isa/<file>.s (kernel-resource-usage remarks on stderr).  `awk '/^<mangled>:/,/s_endpgm/' /tmp/isa/<file>.s` shows one kernel.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
f=$1; shift
mkdir -p /tmp/isa
base=$(basename "$f" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -I"$root/include" -I"$root/clover_amd/csrc" "$@" -S --cuda-device-only -o /tmp/isa/$base.s "$root/clover_amd/csrc/$base.hip" \
  -Rpass-analysis=kernel-resource-usage 2> /tmp/isa/$base.remarks || { cat /tmp/isa/$base.remarks | grep -E "error" -A5; exit 1; }
grep -E "Function Name|VGPRs:|Occupancy|ScratchSize" /tmp/isa/$base.remarks | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - | sed 's/Function Name: //'
