#!/bin/bash
# Round-4 evidence (run on the GPU box through gpurun); parts: a | b
#   a: bench line + kernel stats + HBM traffic passes + NTT launch table (tools/collect_profiles.sh r04)
#   b: SQ counter passes of the transform (VALU busy) and of the element-wise kernels, launch table of a one-stream trace
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
part=${1:-a}
if [ "$part" = a ]; then
  bash tools/collect_profiles.sh r04 || exit 1
  python3 tools/ntt_launch_table.py $O/stats > $O/ntt_launch_table.txt 2>&1 || exit 1
fi
if [ "$part" = b ]; then
  for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
    d=$O/ntt_pmc_$(echo $pass | cut -c4-8)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -- python3 tools/ntt_prof_one.py 56 > $d.log 2>&1 || exit 1
  done
  python3 tools/ntt_pmc_summary.py $O/ntt_pmc_* > $O/ntt_pmc.txt 2>&1 || exit 1
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/ew_pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-profile > $O/ew_pmc.log 2>&1 || exit 1
  python3 tools/elementwise_pmc.py $O/ew_pmc $O/pmc_fetch $O/pmc_write > $O/elementwise_roofline.txt 2>&1 || exit 1
  APSU_HE_SPLIT=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_one_stream -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-profile > $O/trace_one_stream.log 2>&1 || exit 1
  python3 tools/ntt_launch_table.py $O/trace_one_stream > $O/ntt_launch_table_one_stream.txt 2>&1 || exit 1
fi
echo done $part
