#!/bin/bash
# compile one translation unit of csrc/ for gfx950 and print every kernel's register / scratch / LDS use
# usage: tools/kres.sh conv_h2k.hip [filter]
set -e
cd "$(dirname "$0")/../object_detection_tracking_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I . -c "$1" -o /tmp/kres_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|ScratchSize|VGPRs:|Spill|LDS Size" \
  | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//; s/Function Name: //' | paste - - - - - - | grep -E "${2:-.}" | cut -c1-260
rm -f /tmp/kres_$$.o
