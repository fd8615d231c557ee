#!/bin/bash
# Memory-side counters of the join (L1/L2 latencies, hit rates, TLB), four --pmc passes of at most four counters per
# hardware block (more: "exceeds the capabilities of the hardware", and the aborted profiler hangs — hence the timeouts).
# usage (through gpurun): bash tools/pmc_memory_join.sh <tag>
set -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
args="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scale-set --no-host-path --no-traffic"
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCC_REQ_sum TCC_TAG_STALL_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout -k 5 150 rocprofv3 --pmc $set --kernel-trace -d $out/pm$i --output-format csv -- python3 $args > $out/pm$i.log 2>&1 || { grep -m3 "error\|exceeds" $out/pm$i.log; echo "pass $i failed"; rm -rf $out/pm$i; continue; }
    python3 tools/pmc_summary.py $out/pm$i k_join_lds > $out/${tag}_pmc_mem${i}_join.json
    rm -rf $out/pm$i
    echo "pass $i done"
done
cat $out/${tag}_pmc_mem*_join.json
