#!/bin/bash
set -o pipefail
O=gpurun_out/r03_exp12
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_path.py -x -q -m gpu -k "32768 or FAMILIES or family or every_prime_width or ring_size" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
