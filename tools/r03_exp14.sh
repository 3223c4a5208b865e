#!/bin/bash
set -o pipefail
O=gpurun_out/r03_exp14
mkdir -p $O
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_EVAL_WS_BYTES=6442450944 --b APSU_HE_EVAL_WS_BYTES=3300000000 > $O/ab_ws2.log 2>&1 || { tail -20 $O/ab_ws2.log; exit 1; }
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_EVAL_WS_BYTES=6442450944 --b APSU_HE_EVAL_WS_BYTES=1700000000 > $O/ab_ws4.log 2>&1 || exit 1
grep -h "B - A" $O/ab_*.log
