#!/bin/bash
# tools/ab_variants.sh — VARIANTS="orig NAME …": the default bench with the shipped library (orig) and with the variants
# tools/build_variant.sh made, back to back on one GPU box; one line per run: step ms, join ms (roofline), join ms (stage), rows in the filter tier
mkdir -p gpurun_out/r04b
cp pandelos_amd/lib/libpandelos_amd.so /tmp/orig.so
trap 'cp /tmp/orig.so pandelos_amd/lib/libpandelos_amd.so' EXIT      # (whatever happens, the shipped library comes back)
for v in ${VARIANTS:-orig}; do
  if [ $v != orig ]; then cp pandelos_amd/lib/variants/libpandelos_amd_$v.so pandelos_amd/lib/libpandelos_amd.so; else cp /tmp/orig.so pandelos_amd/lib/libpandelos_amd.so; fi
  timeout -k 10 200 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r04b/$v.json 2>/dev/null || exit 1
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/r04b/$v.json').read().strip().splitlines()[-1]);print('$v',round(d['ms_per_step'],4),round(d['roofline']['launch_ms'],4),round(d['stage_ms']['join'],4),d['stage_ms']['tier1_rows'])"
done
cp /tmp/orig.so pandelos_amd/lib/libpandelos_amd.so
