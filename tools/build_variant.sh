#!/bin/bash
# Builds a variant of the kernel library with extra compiler flags, for experiments:
#   tools/build_variant.sh <name> <flags...>   ->  diffusion_model_nemo_amd/lib/libdmn_hip_<name>.so
# (tools that take DMN_LIB_VARIANT=<name> load it instead of the product library)
set -e
name=$1; shift
cd "$(dirname "$0")/../diffusion_model_nemo_amd/csrc"
obj=/tmp/dmn_variant_$name; mkdir -p $obj
common="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -fno-honor-nans"
# the product build also passes: -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops (add them here to keep them)
pids=()
for f in *.hip *.cpp; do
  extra=""
  case $f in
    dmn_linattn.hip|dmn_attn.hip) extra="-mllvm -amdgpu-mfma-vgpr-form";;
    dmn_train_gemm.hip|dmn_train_ops.hip) extra="-munsafe-fp-atomics";;
  esac
  hipcc $common $extra "$@" -x hip -c $f -o $obj/${f%.*}.o &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libdmn_hip_$name.so $obj/*.o
echo built ../lib/libdmn_hip_$name.so
