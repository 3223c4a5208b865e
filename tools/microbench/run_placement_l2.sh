#!/bin/bash
# L2 hit / miss and memory-side requests per k_mac dispatch on four identical copies of the database, several processes (is a slow copy one
# whose rows evict more of the shared powers from L2?)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05/place_l2
mkdir -p $O
cd tools/microbench/_bin
for r in 1 2 3 4; do
  d=$O/p$r; mkdir -p $d
  PLACEMENT=4 timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $d -- ./macbench_place 1 > $d.log 2>&1 || echo "run $r failed"
  echo "== process $r"; grep "^pass 2" $d.log
  python3 ../../pmc_by_dispatch.py $d "k_mac<" 1000000 | tail -34 | awk 'NR<=2 || NR%8==3'
done
