#!/bin/bash
set -o pipefail
O=gpurun_out/r03_exp13
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -14 $O/pytest.log
