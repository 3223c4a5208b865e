#!/bin/bash
set -o pipefail
O=gpurun_out/r03_exp3
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 500 python tools/shard_probe.py --worlds 1,8 --splits 0,1 --asyncs 1 > $O/probe.log 2>&1 || { tail -20 $O/probe.log; exit 1; }
APSU_HE_MAX_INFLIGHT=1 timeout -k 10 500 python tools/shard_probe.py --worlds 1,8 --splits 1 --asyncs 1 > $O/probe_inflight1.log 2>&1 || exit 1
APSU_HE_MAX_INFLIGHT=4 timeout -k 10 500 python tools/shard_probe.py --worlds 1,8 --splits 1 --asyncs 1 > $O/probe_inflight4.log 2>&1 || exit 1
APSU_HE_MAX_INFLIGHT=64 timeout -k 10 500 python tools/shard_probe.py --worlds 1,8 --splits 1 --asyncs 1 > $O/probe_inflight64.log 2>&1 || exit 1
grep -h "^world\|events" $O/probe*.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cat $O/bench.json
