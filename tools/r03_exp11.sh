#!/bin/bash
set -o pipefail
O=gpurun_out/r03_exp11
mkdir -p $O
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_NTT_PRIO=0 --b APSU_HE_NTT_PRIO=1 > $O/ab_prio.log 2>&1 || { tail -20 $O/ab_prio.log; exit 1; }
grep -h "B - A" $O/ab_*.log
