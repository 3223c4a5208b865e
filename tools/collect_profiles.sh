#!/bin/bash
# Run ON THE GPU BOX (through gpurun) to regenerate the round's evidence under gpurun_out/; copy into profiles/ afterwards.
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r01'
set -o pipefail
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench.json || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-host-io > $OUT/stats.log 2>&1 || exit 1
grep -v "^W\|^E\|^I\|amdgpu.ids" $OUT/stats.log | tail -1 > $OUT/bench_under_rocprof.json
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-profile > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-profile > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/ntt_stream_bench.py > $OUT/ntt_stream.txt 2>&1 || exit 1
echo collected $OUT
