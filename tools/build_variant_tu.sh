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
if [ $tu = kt_bulk ]; then  # the in-flight-register checks of csrc/Makefile, reported (a variant is for timing only when they fail)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. "$@" -Rpass-analysis=kernel-resource-usage -x hip --cuda-device-only -S $tu.hip -o build/${tu}_$name.s 2> build/${tu}_$name.remarks
  awk '/Function Name:/ {f = $0} /ScratchSize/ && f ~ /part2_swwc_kernel|part2_fast_kernel/ && $0 !~ /lane\]: 0 / {print "  SCRATCH: " $0}' build/${tu}_$name.remarks
  python3 ../../tools/check_inflight.py build/${tu}_$name.s "part2_swwc_kernel|part2_fast_kernel" | tail -2 | cut -c1-200
fi
objs=""
for o in kt_host kt_oligo kt_oligo_generic kt_ctr kt_bulk kt_shard kt_cov kt_cgr kt_min kt_synth; do
  if [ $o = $tu ]; then objs="$objs build/${tu}_$name.o"; else objs="$objs build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib$name.so $objs -ldl
echo built ../variants/lib$name.so
