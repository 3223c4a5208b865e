#!/bin/bash
# why does k_mac run 10 % faster on one copy of the database than on another in the same process?  counters per dispatch
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05/place_pmc
mkdir -p $O
cd tools/microbench/_bin
i=0
for pass in \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum GRBM_GUI_ACTIVE" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1)); d=$O/p$i; mkdir -p $d
  PLACEMENT=4 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -- ./macbench_place 1 > $d.log 2>&1 || echo "pass $i failed: $(tail -1 $d.log)"
  grep "^pass 2" $d.log
  python3 ../../pmc_by_dispatch.py $d "k_mac<" 1000000 | tail -16
done
