#!/bin/bash
# Round-6 evidence (run on the GPU box through gpurun); parts:
#   base:    the tree as round 5 left it -- lone-limb transform latency (16M and 1M rings), the 1M-1024-com query's launch gaps and timeline,
#            the 256M-4096 per-rank cost table (N = 1, 2, 4, 8 shards executed on one GPU)
#   pmc:     the counter passes round 5 left open: TA / TD of the in-path k_mac launch in passes of <= 3 block counters (round 5's pass 7 asked
#            for five + GRBM_GUI_ACTIVE: "error code 38: request exceeds the capabilities of the hardware to collect"), the in-path forward
#            transform and the staged RAW inverse (6 792-limb launch)
#   place:   k_mac on a slow and a fast copy of the same database (tools/microbench/macbench.hip PLACEMENT mode): memory-side request counters
#            PER TCC INSTANCE (json output keeps the dimensions), not summed
#   after:   the same measurements as `base` on the current tree + the 16M rank-cost table
#   bench:   bench line + rocprofv3 kernel stats of the same command (tools/collect_profiles.sh r06) + one-stream launch table
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
part=${1:-base}
tag=${2:-$part}
if [ "$part" = base ] || [ "$part" = after ]; then
  timeout -k 10 200 python3 tools/ntt_latency.py 16M-4096 > $O/ntt_latency_16M_$tag.txt 2>&1 || { echo "ntt_latency 16M failed"; tail -3 $O/ntt_latency_16M_$tag.txt; exit 1; }
  timeout -k 10 200 python3 tools/ntt_latency.py 1M-1024-com > $O/ntt_latency_1M_$tag.txt 2>&1 || { echo "ntt_latency 1M failed"; tail -3 $O/ntt_latency_1M_$tag.txt; exit 1; }
  echo "latency tables done"
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_1M_$tag -- python3 bench.py --config 1M-1024-com --steps 6 --warmup 2 --no-cpu-baseline --no-host-io --no-profile > $O/trace_1M_$tag.log 2>&1 || { echo "1M trace failed"; tail -3 $O/trace_1M_$tag.log; exit 1; }
  python3 tools/launch_gaps.py $O/trace_1M_$tag > $O/launch_gaps_1M_$tag.txt 2>&1
  python3 tools/query_timeline.py $O/trace_1M_$tag > $O/query_timeline_1M_$tag.txt 2>&1
  rm -rf $O/trace_1M_$tag
  timeout -k 10 200 python3 bench.py --config 1M-1024-com --no-cpu-baseline > $O/bench_1M_$tag.json 2> $O/bench_1M_$tag.err || { echo "1M bench failed"; tail -3 $O/bench_1M_$tag.err; exit 1; }
  echo "1M done"
  if [ "$part" = after ]; then
    RANK_COST_REPEAT=3 timeout -k 10 400 python3 tools/rank_cost.py 16M-4096 > $O/rank_cost_16M_$tag.txt 2>&1 || { echo "rank_cost 16M failed"; tail -3 $O/rank_cost_16M_$tag.txt; exit 1; }
  fi
  RANK_COST_REPEAT=3 timeout -k 10 900 python3 tools/rank_cost.py 256M-4096 > $O/rank_cost_256M_$tag.txt 2>&1 || { echo "rank_cost 256M failed"; tail -3 $O/rank_cost_256M_$tag.txt; exit 1; }
  tail -4 $O/rank_cost_256M_$tag.txt
fi
if [ "$part" = pmc ]; then
  i=0
  for pass in \
    "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
    "TD_TD_BUSY_sum TD_TC_STALL_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LEVEL_WAVES GRBM_GUI_ACTIVE" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
    "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
    "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" \
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" ; do
    i=$((i+1))
    d=$O/pmc/p$(printf %02d $i)
    mkdir -p $d
    if ! timeout -k 10 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -- python3 tools/mac_prof_one.py 3 > $d.log 2>&1; then
      echo "pass $i FAILED ($pass): $(tail -2 $d.log | tr '\n' ' ')"
    else
      echo "pass $i ok"
    fi
  done
  python3 tools/mac_pmc_summary.py $O/pmc > $O/mac_pmc_ta_td.txt 2>&1
  python3 tools/mac_pmc_summary.py $O/pmc "k_ntt<13, false" > $O/ntt_fwd_pmc_inpath.txt 2>&1
  python3 tools/mac_pmc_summary.py $O/pmc "k_ntt<13, true" > $O/ntt_inv_staged_pmc_inpath.txt 2>&1
  python3 tools/mac_pmc_summary.py $O/pmc "k_intt_tensor<13" > $O/ntt_tensor_pmc_inpath.txt 2>&1
  tail -30 $O/mac_pmc_ta_td.txt
fi
if [ "$part" = pmc_ta ]; then
  # the TA block holds TWO counters per pass on gfx950: three (pass 1 of `pmc`) are refused like round 5's five ("error code 38")
  i=7
  for pass in \
    "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
    "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE" ; do
    i=$((i+1))
    d=$O/pmc/p$(printf %02d $i)
    mkdir -p $d
    if ! timeout -k 10 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -- python3 tools/mac_prof_one.py 3 > $d.log 2>&1; then
      echo "pass $i FAILED ($pass): $(grep -v '^W\|^I\|^    @' $d.log | tail -2 | tr '\n' ' ')"
    else
      echo "pass $i ok"
    fi
  done
fi
if [ "$part" = ab ]; then
  # in-process A/B of the transform forms: every launch in the 16-coefficient form against the default selection
  for spec in "16M-4096 1" "16M-4096 8" "16M-4096 4" "1M-1024-com 1" "256M-4096 8"; do
    set -- $spec
    timeout -k 10 500 python3 tools/ab_compare.py --a APSU_HE_NTT_LATENCY_LIMBS=0 --b "" --config $1 --world $2 >> $O/ab_ntt_forms_$tag.txt 2>&1 || { echo "ab $spec failed"; tail -3 $O/ab_ntt_forms_$tag.txt; exit 1; }
  done
  grep -v amdgpu.ids $O/ab_ntt_forms_$tag.txt
fi
if [ "$part" = place ]; then
  cd tools/microbench/_bin || exit 1
  OO=../../../$O/place
  mkdir -p $OO
  i=0
  for pass in \
    "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL GRBM_GUI_ACTIVE" \
    "TCC_REQ TCC_HIT TCC_MISS GRBM_GUI_ACTIVE" \
    "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_BUSY GRBM_GUI_ACTIVE" ; do
    i=$((i+1)); d=$OO/p$i; mkdir -p $d
    PLACEMENT=4 timeout -k 10 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv json -d $d -- ./macbench_place 1 > $d.log 2>&1 || echo "pass $i failed: $(tail -1 $d.log)"
    grep "^pass 2" $d.log
  done
  cd ../../..
  python3 tools/pmc_by_instance.py $O/place "k_mac<" > $O/mac_placement_by_instance.txt 2>&1
  tail -40 $O/mac_placement_by_instance.txt
  find $O/place -name "*.json" -size +20M -delete
fi
if [ "$part" = place2 ]; then
  # second look at the placement effect, on a box where copies differ: where do the memory-side reads go (local DRAM / GMI / IO), per TCC instance
  cd tools/microbench/_bin || exit 1
  OO=../../../$O/place2
  mkdir -p $OO
  i=0
  for pass in \
    "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL GRBM_GUI_ACTIVE" \
    "TCC_EA0_RDREQ_DRAM TCC_EA0_RDREQ_DRAM_32B TCC_EA0_RDREQ_GMI_32B TCC_EA0_RDREQ_IO_32B GRBM_GUI_ACTIVE" \
    "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL GRBM_GUI_ACTIVE" \
    "TCC_EA0_RDREQ_DRAM TCC_EA0_RDREQ_DRAM_32B TCC_EA0_RDREQ_GMI_32B TCC_EA0_RDREQ_IO_32B GRBM_GUI_ACTIVE" \
    "TCC_HIT TCC_MISS TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_GMI_CREDIT_STALL GRBM_GUI_ACTIVE" \
    "TCC_HIT TCC_MISS TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_GMI_CREDIT_STALL GRBM_GUI_ACTIVE" ; do
    i=$((i+1)); d=$OO/p$i; mkdir -p $d
    PLACEMENT=4 timeout -k 10 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv json -d $d -- ./macbench_place 1 > $d.log 2>&1 || echo "pass $i failed: $(grep -v '^W\|^I\|^    @' $d.log | tail -2 | tr '\n' ' ')"
    grep "^pass 2" $d.log
  done
  cd ../../..
  python3 tools/pmc_by_instance.py $O/place2 "k_mac<" 32 > $O/mac_placement2_by_instance.txt 2>&1
  find $O/place2 -name "*.json" -delete
  grep -c dispatch $O/mac_placement2_by_instance.txt
fi
if [ "$part" = bench ]; then
  bash tools/collect_profiles.sh r06 || exit 1
  APSU_HE_SPLIT=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_one_stream -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-io --no-profile > $O/trace_one_stream.log 2>&1 || exit 1
  python3 tools/ntt_launch_table.py $O/trace_one_stream > $O/ntt_launch_table_one_stream.txt 2>&1 || exit 1
  rm -rf $O/trace_one_stream
fi
echo done $part
