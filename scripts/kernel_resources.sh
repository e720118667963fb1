#!/bin/bash
# scripts/kernel_resources.sh [extra hipcc flags]: registers / LDS / scratch of the kernels BASELINE configs[2]
# launches (tuning build, KAS_MINIMAL_INSTANCES), from the compiler's own resource-usage remarks.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DKAS_MINIMAL_INSTANCES "$@" \
  -Rpass-analysis=kernel-resource-usage -I"$ROOT/include" -I"$ROOT/kafka-assigner_amd/csrc" -o /dev/null \
  "$ROOT/kafka-assigner_amd/csrc/kas_hip.hip" 2>&1 | grep -E "remark:" | sed -e 's/.*remark: //' | \
  grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" | paste - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | tr -s ' \t' ' '
