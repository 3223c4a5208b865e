#!/bin/bash
set -o pipefail
O=gpurun_out/r03_exp10
mkdir -p $O
APSU_HE_EXT_V2=1 timeout -k 10 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_params_sweep.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_EXT_V2=0 --b APSU_HE_EXT_V2=1 > $O/ab_extv2.log 2>&1 || { tail -20 $O/ab_extv2.log; exit 1; }
grep -h "B - A" $O/ab_*.log
