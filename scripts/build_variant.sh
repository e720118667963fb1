#!/bin/bash
# scripts/build_variant.sh NAME [PATCH ...] [-- extra hipcc flags]
# Tuning build of the C-ABI library: copies kafka-assigner_amd/csrc to a scratch directory, applies
# the given patches (-p0 paths as in experiments/*.patch), compiles only the kernels BASELINE
# configs[2] launches (KAS_MINIMAL_INSTANCES: seconds instead of minutes; pass
# -DKAS_MINIMAL_INSTANCES=5 among the extra flags for the configs[4] kernels instead) into
# variants/libkas_hip_NAME.so.  Select it with KAS_HIP_LIB=variants/libkas_hip_NAME.so.
# variants/*.so is git-ignored but travels to the GPU box.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
PATCHES=(); FLAGS=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; FLAGS=("$@"); break; fi
  PATCHES+=("$1"); shift
done
MIN=-DKAS_MINIMAL_INSTANCES
for f in "${FLAGS[@]}"; do case "$f" in -DKAS_MINIMAL_INSTANCES=*) MIN="";; esac; done
W=$(mktemp -d /tmp/kasvar.XXXXXX)
mkdir -p "$W/kafka-assigner_amd" "$ROOT/variants"
cp -r "$ROOT/kafka-assigner_amd/csrc" "$W/kafka-assigner_amd/csrc"
rm -f "$W"/kafka-assigner_amd/csrc/*.so
for p in "${PATCHES[@]}"; do (cd "$W" && patch -p0 -s < "$ROOT/$p"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
  $MIN "${FLAGS[@]}" -I"$ROOT/include" -I"$W/kafka-assigner_amd/csrc" \
  -o "$ROOT/variants/libkas_hip_$NAME.so" "$W/kafka-assigner_amd/csrc/kas_hip.hip"
rm -rf "$W"
echo "variants/libkas_hip_$NAME.so"
