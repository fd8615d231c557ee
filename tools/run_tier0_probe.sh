export TMPDIR=/tmp; out=gpurun_out/$1; mkdir -p $out; A="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scale-set --no-host-path --no-traffic"
(timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "partition" > $out/t1.log 2>&1; echo "rc=$?" >> $out/t1.log); tail -3 $out/t1.log
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-scale-set --no-host-path --no-traffic > $out/bench_t0.json 2> $out/bench_t0.err
timeout -k 5 100 rocprofv3 --kernel-trace --stats -d $out/ks --output-format csv -- python3 $A > $out/ks.log 2>&1; cp "$(find $out/ks -name '*kernel_stats.csv' | sort | tail -1)" $out/kernel_stats.csv; rm -rf $out/ks; grep -E "k_join|k_order|k_mirror" $out/kernel_stats.csv | cut -c1-60,60-200 | awk -F, '{print $1, $2, $4}'
timeout -k 5 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d $out/pb --output-format csv -- python3 $A > $out/pb.log 2>&1; python3 tools/pmc_summary.py $out/pb k_join_part > $out/sq_insts.json; rm -rf $out/pb
timeout -k 5 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace -d $out/pa --output-format csv -- python3 $A > $out/pa.log 2>&1; python3 tools/pmc_summary.py $out/pa k_join_part > $out/sq_cycles.json; rm -rf $out/pa
python3 -c "
import json
d=json.load(open('$out/bench_t0.json')); print(d['ms_per_step'], d['stage_ms']['join'], d['roofline']['frac'])
a=json.load(open('$out/sq_insts.json'))['k_join_part']; b=json.load(open('$out/sq_cycles.json'))['k_join_part']
print({k:round(v/1e6,1) for k,v in a.items()}); print({k:round(v/1e6,1) for k,v in b.items()})"
