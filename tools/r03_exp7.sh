#!/bin/bash
set -o pipefail
O=gpurun_out/r03_exp7
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python tools/multi_bench.py --devices "0;0,0" --steps 20 > $O/multi.log 2>&1 || { tail -20 $O/multi.log; exit 1; }
grep -v "amdgpu.ids\|version\|Hostname\|Librccl" $O/multi.log
