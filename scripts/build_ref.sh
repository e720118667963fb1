#!/bin/bash
# scripts/build_ref.sh GITREF NAME [-- extra hipcc flags]: the C-ABI library of another commit (csrc/ + include/ as of GITREF)
# into variants/libkas_hip_NAME.so, to A/B the working tree against on one GPU box (tools/ab_harness, scripts/trip_benchab.sh).
# Default: tuning build (KAS_MINIMAL_INSTANCES, the kernels BASELINE configs[2] launches: seconds); pass
# -- -DKAS_MINIMAL_INSTANCES=0 for every instance.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=$1; NAME=$2; shift 2
FLAGS=()
if [ "${1:-}" == "--" ]; then shift; FLAGS=("$@"); fi
MIN=-DKAS_MINIMAL_INSTANCES
for f in "${FLAGS[@]}"; do case "$f" in -DKAS_MINIMAL_INSTANCES=*) MIN="";; esac; done
W=$(mktemp -d /tmp/kasref.XXXXXX)
mkdir -p "$ROOT/variants"
(cd "$ROOT" && git archive "$REF" kafka-assigner_amd/csrc include) | tar -x -C "$W"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
  $MIN "${FLAGS[@]}" -I"$W/include" -I"$W/kafka-assigner_amd/csrc" \
  -o "$ROOT/variants/libkas_hip_$NAME.so" "$W/kafka-assigner_amd/csrc/kas_hip.hip"
rm -rf "$W"
echo "variants/libkas_hip_$NAME.so"
