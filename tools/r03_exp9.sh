#!/bin/bash
set -o pipefail
O=gpurun_out/r03_exp9
mkdir -p $O
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_HIGH_LATE=0 --b APSU_HE_HIGH_LATE=1 > $O/ab_late.log 2>&1 || { tail -20 $O/ab_late.log; exit 1; }
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_HIGH_LATE=0 --b APSU_HE_HIGH_LATE=1 --world 8 --steps 30 > $O/ab_late8.log 2>&1 || exit 1
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_MAX_INFLIGHT=2 --b APSU_HE_MAX_INFLIGHT=4 > $O/ab_inflight.log 2>&1 || exit 1
grep -h "B - A" $O/ab_*.log
