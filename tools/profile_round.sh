#!/bin/bash
# Everything profiles/ holds for a round, on the GPU box: bench line, kernel stats, SQ counters and HBM traffic of the join.
# usage (through gpurun): bash tools/profile_round.sh <tag>      -> files under gpurun_out/<tag>_*
set -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
python3 bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err || exit 1
rocprofv3 --kernel-trace --stats -d $out/ks --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-scale-set --no-host-path --no-traffic > $out/ks.log 2>&1 || exit 1
cp "$(find $out/ks -name '*kernel_stats.csv' | sort | tail -1)" $out/${tag}_bench_default_kernel_stats.csv; rm -rf $out/ks
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pf --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scale-set --no-host-path --no-traffic > $out/pf.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pw --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scale-set --no-host-path --no-traffic > $out/pw.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $out/pf $out/pw mycoplasma64_standin "$tag" > $out/${tag}b_pmc_traffic_join.json; rm -rf $out/pf $out/pw
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace -d $out/pa --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scale-set --no-host-path --no-traffic > $out/pa.log 2>&1 || exit 1
python3 tools/pmc_summary.py $out/pa k_join > $out/${tag}_pmc_sq_cycles_join.json; rm -rf $out/pa
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d $out/pb --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scale-set --no-host-path --no-traffic > $out/pb.log 2>&1 || exit 1
python3 tools/pmc_summary.py $out/pb k_join > $out/${tag}_pmc_sq_insts_join.json; rm -rf $out/pb
tail -c 600 $out/${tag}_bench_default.json
