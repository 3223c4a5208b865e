#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_exp8
mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-io > $O/stats.log 2>&1 || { tail $O/stats.log; exit 1; }
grep -v "^W\|^E\|^I\|amdgpu.ids" $O/stats.log | tail -1 > $O/bench_under_rocprof.json
python3 tools/ntt_launch_table.py $(ls -t $O/stats/*/*kernel_trace.csv | head -1) > $O/launch_table.txt 2>&1
cat $O/launch_table.txt | tail -45
