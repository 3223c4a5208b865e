#!/bin/bash
# Round-5 evidence (run on the GPU box through gpurun); parts: mac | bench | trace
#   mac:   what the waves of the in-path k_mac launch wait on -- SQ / TCP / TA / TD / TCC counter passes of tools/mac_prof_one.py
#   bench: bench line + rocprofv3 kernel stats of the same command (tools/collect_profiles.sh r05)
#   trace: kernel trace of a one-stream run -> NTT launch table
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
part=${1:-mac}
if [ "$part" = mac ]; then
  i=0
  for pass in \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LEVEL_WAVES GRBM_GUI_ACTIVE" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
    "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
    "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
    "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" \
    "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
    "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" \
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE" \
    "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE" \
    "TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" \
    "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_IB_STALL_sum GRBM_GUI_ACTIVE" ; do
    i=$((i+1))
    d=$O/mac_pmc/p$(printf %02d $i)
    mkdir -p $d
    if ! timeout -k 10 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -- python3 tools/mac_prof_one.py 3 > $d.log 2>&1; then
      echo "pass $i FAILED ($pass): $(tail -2 $d.log | tr '\n' ' ')"
    else
      echo "pass $i ok"
    fi
  done
  python3 tools/mac_pmc_summary.py $O/mac_pmc > $O/mac_pmc.txt 2>&1
  python3 tools/mac_pmc_summary.py $O/mac_pmc "k_ntt<13, false" > $O/ntt_fwd_pmc_inpath.txt 2>&1
  tail -60 $O/mac_pmc.txt
fi
if [ "$part" = bench ]; then
  bash tools/collect_profiles.sh r05 || exit 1
fi
if [ "$part" = trace ]; then
  APSU_HE_SPLIT=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_one_stream -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-profile > $O/trace_one_stream.log 2>&1 || exit 1
  python3 tools/ntt_launch_table.py $O/trace_one_stream > $O/ntt_launch_table_one_stream.txt 2>&1 || exit 1
fi
echo done $part
