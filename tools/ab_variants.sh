mkdir -p gpurun_out/r04b
cp pandelos_amd/lib/libpandelos_amd.so /tmp/orig.so
for v in ${VARIANTS:-orig}; do
  if [ $v != orig ]; then cp pandelos_amd/lib/variants/libpandelos_amd_$v.so pandelos_amd/lib/libpandelos_amd.so; else cp /tmp/orig.so pandelos_amd/lib/libpandelos_amd.so; fi
  timeout -k 10 200 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r04b/$v.json 2>/dev/null || exit 1
  python3 -c "
import json,sys;d=json.loads(open('gpurun_out/r04b/$v.json').read().strip().splitlines()[-1]);print('$v',round(d['ms_per_step'],4),round(d['roofline']['launch_ms'],4),round(d['stage_ms']['join'],4),d['stage_ms']['tier1_rows'])"
done
cp /tmp/orig.so pandelos_amd/lib/libpandelos_amd.so
