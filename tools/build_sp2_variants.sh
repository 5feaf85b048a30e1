#!/bin/bash
# tuning builds of the bf16x6 v2 kernel (conv_gemm_split.hip): the plain-epilogue instantiation only, one library per experiment switch.
#   bash tools/build_sp2_variants.sh PROF "PROF X_NOMFMA" ...      -> xfr_amd/csrc/variants/libxfr_amd_<names>.so
cd "$(dirname "$0")/../xfr_amd/csrc" || exit 1
mkdir -p variants
pids=()
for v in "$@"; do
  name=$(echo "$v" | tr ' ' '_')
  defs="-DSP2_ONLY_PLAIN"
  for d in $v; do defs="$defs -DSP2_$d"; done
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $defs -c conv_gemm_split.hip -o variants/split_$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libxfr_amd_$name.so engine.o conv_gemm.o variants/split_$name.o elementwise.o saliency.o -ldl &&
    rm -f variants/split_$name.o && echo "built $name" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
