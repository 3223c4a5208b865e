#!/bin/bash
# Round-3 evidence beyond tools/collect_profiles.sh (run on the GPU box through gpurun); parts: a | b | c
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
part=${1:-a}
if [ "$part" = a ]; then
  bash tools/collect_profiles.sh r03 || exit 1
  python3 tools/ntt_launch_table.py $O/stats > $O/ntt_launch_table.txt 2>&1 || exit 1
fi
if [ "$part" = b ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_n1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-profile > $O/trace_n1.log 2>&1 || exit 1
  python3 tools/launch_gaps.py $O/trace_n1 > $O/launch_gaps_n1.txt 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_n8 -- python3 tools/trace_small.py > $O/trace_n8.log 2>&1 || exit 1
  python3 tools/launch_gaps.py $O/trace_n8 > $O/launch_gaps_n8.txt 2>&1 || exit 1
  RANK_COST_REPEAT=10 timeout -k 10 600 python3 tools/rank_cost.py > $O/rank_cost.txt 2>&1 || exit 1
  timeout -k 10 500 python3 tools/shard_probe.py --worlds 1,8 --splits 0,1 --asyncs 0,1 > $O/shard_probe.txt 2>&1 || exit 1
  timeout -k 10 400 python3 tools/multi_bench.py --devices "0;0,0" --steps 20 > $O/multi_bench.txt 2>&1 || exit 1
  for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
    d=$O/ntt_pmc_$(echo $pass | cut -c4-8)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -- python3 tools/ntt_prof_one.py 56 > $d.log 2>&1 || exit 1
  done
  python3 tools/ntt_pmc_summary.py $O/ntt_pmc_* > $O/ntt_pmc.txt 2>&1 || exit 1
fi
if [ "$part" = c ]; then
  timeout -k 10 500 python3 bench.py --config 1M-1024-com --steps 20 --warmup 5 --no-host-io 2>&1 | tail -1 > $O/bench_1M.json || exit 1
  timeout -k 10 900 python3 bench.py --config 256M-4096 --steps 10 --warmup 3 --no-cpu-baseline --no-host-io 2>&1 | tail -1 > $O/bench_256M.json || exit 1
  RANK_COST_REPEAT=3 timeout -k 10 900 python3 tools/rank_cost.py 256M-4096 > $O/rank_cost_256M.txt 2>&1 || exit 1
  timeout -k 10 600 python3 tests/full_query_parity.py 16M-4096 > $O/full_query_parity.txt 2>&1 || exit 1
  timeout -k 10 300 python3 tests/full_query_parity.py 1M-1024-com >> $O/full_query_parity.txt 2>&1 || exit 1
fi
echo done $part
