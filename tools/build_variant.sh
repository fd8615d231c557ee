#!/bin/bash
# tools/build_variant.sh NAME "-DMACRO ..." : libpandelos_amd.so with pdl_join.hip compiled under extra flags ->
# pandelos_amd/lib/variants/libpandelos_amd_NAME.so (kernel experiments: copy it over lib/libpandelos_amd.so on the GPU box)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p pandelos_amd/lib/variants /tmp/pdl_variant_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $flags -c pandelos_amd/csrc/pdl_join.hip -o /tmp/pdl_variant_$name/pdl_join.o
objs=""
for o in pdl_sort pdl_dict pdl_bbh pdl_ingest pdl_api; do objs="$objs pandelos_amd/lib/obj/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pandelos_amd/lib/variants/libpandelos_amd_$name.so $objs /tmp/pdl_variant_$name/pdl_join.o
echo built pandelos_amd/lib/variants/libpandelos_amd_$name.so
