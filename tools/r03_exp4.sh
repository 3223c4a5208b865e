#!/bin/bash
set -o pipefail
O=gpurun_out/r03_exp4
mkdir -p $O
APSU_HE_FUSE_KS=1 timeout -k 10 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_ops.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_FUSE_KS=0 --b APSU_HE_FUSE_KS=1 > $O/ab_ks.log 2>&1 || { tail -20 $O/ab_ks.log; exit 1; }
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_FUSE_KS=0 --b APSU_HE_FUSE_KS=1 --world 8 --steps 30 > $O/ab_ks8.log 2>&1 || exit 1
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_FUSE_EXT=0 --b APSU_HE_FUSE_EXT=1 > $O/ab_ext.log 2>&1 || exit 1
timeout -k 10 300 python tools/ab_test.py --a APSU_HE_SPLIT=0 --b APSU_HE_SPLIT=1 > $O/ab_split.log 2>&1 || exit 1
grep -h "B - A" $O/ab_*.log
