#!/bin/bash
# usage: tools/build_variant_tu.sh <name> <tu> [extra hipcc flags]
# builds kmertools_amd/variants/lib<name>.so with translation unit <tu> (e.g. kt_cov) recompiled
# with extra flags (A/B timing of -D variants in one gpurun call via KT_LIB)
set -e
cd "$(dirname "$0")/../kmertools_amd/csrc"
name=$1; tu=$2; shift 2
mkdir -p ../variants build
make -s >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -I. "$@" -x hip -c $tu.hip -o build/${tu}_$name.o
objs=""
for o in kt_host kt_oligo kt_oligo_generic kt_ctr kt_bulk kt_shard kt_cov kt_cgr kt_min kt_synth; do
  if [ $o = $tu ]; then objs="$objs build/${tu}_$name.o"; else objs="$objs build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib$name.so $objs -ldl
echo built ../variants/lib$name.so
