#!/bin/bash
# address-translation counters per k_mac dispatch on four identical copies of the database, several processes
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05/place_tlb
mkdir -p $O
cd tools/microbench/_bin
i=0
for pass in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
            "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
            "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
            "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum"; do
  i=$((i+1)); d=$O/p$i; mkdir -p $d
  PLACEMENT=4 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -- ./macbench_place 1 > $d.log 2>&1 || echo "run $i failed"
  echo "== process $i: $pass"; grep "^pass 2" $d.log
  python3 ../../pmc_by_dispatch.py $d "k_mac<" 1000000 | tail -34 | awk 'NR<=2 || NR%8==3'
done
