#!/bin/bash
# usage: tools/build_variant.sh <name> <kt_oligo.hip source> [extra hipcc flags]
# builds kmertools_amd/variants/lib<name>.so with an alternative kt_oligo.hip (A/B timing in one gpurun call)
set -e
cd "$(dirname "$0")/../kmertools_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../variants build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable --offload-arch=gfx950 -I. "$@" -x hip -c "$src" -o build/oligo_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib$name.so build/kt_host.o build/oligo_$name.o build/kt_ctr.o build/kt_bulk.o build/kt_synth.o
echo built ../variants/lib$name.so
