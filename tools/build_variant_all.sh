#!/bin/bash
# build a variant lib with extra flags applied to ALL translation units
set -e
cd /root/repo/kmertools_amd/csrc
name=$1; shift
mkdir -p ../variants build/v_$name
objs=""
for o in kt_host kt_oligo kt_oligo_generic kt_ctr kt_bulk kt_shard kt_cov kt_cgr kt_min kt_synth; do
  src=$o.hip; [ -f $src ] || src=$o.cpp
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -I. "$@" -x hip -c $src -o build/v_$name/$o.o &
  objs="$objs build/v_$name/$o.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib$name.so $objs -ldl
echo built lib$name.so
