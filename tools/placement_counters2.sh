#!/bin/bash
# Second look at the placement classes: per-instance imbalance (max / min over the TCC channels, per XCD, per channel index) through derived
# counters (tools/placement_extra_counters.yaml), and the two-stream fill probe.  gpurun --timeout 1800 -- 'bash tools/placement_counters2.sh'
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
OUT=$ROOT/gpurun_out/r06_placement
mkdir -p $OUT
export TMPDIR=/tmp
( time tools/_bin/two_stream_probe 10 ) > $OUT/two_stream.log 2>&1
( time tools/_bin/two_stream_probe 10 ) > $OUT/two_stream_b.log 2>&1
i=20
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  d=$OUT/pass$i
  rm -rf $d; mkdir -p $d
  echo "$SET" > $d/counters.txt
  ( cd /tmp && timeout 900 rocprofv3 -E $ROOT/tools/placement_extra_counters.yaml --kernel-trace --pmc $SET -d $d -o run --output-format csv -- python $ROOT/tools/placement_counters.py ) > $d/run.log 2>&1
  echo "pass $i ($SET): rc $?" >> $OUT/passes.log
  for f in $(find $d -name "*counter_collection.csv") $(find $d -name "*kernel_trace.csv"); do
    head -1 $f > $f.small; grep -E "lookup_fill_kernel|trace_kernel" $f >> $f.small; mv $f.small $f
  done
  find $d -name "*agent_info.csv" -delete
done <<'SETS'
H2R_WRREQ_MAX H2R_WRREQ_MIN TCC_EA0_WRREQ_sum
H2R_TCCREQ_MAX H2R_TCCREQ_MIN H2R_TCCBUSY_MAX H2R_TCCBUSY_MIN
H2R_WRSTALL_MAX H2R_WRSTALL_MIN H2R_WRLEVEL_MAX H2R_WRLEVEL_MIN
H2R_TAGSTALL_MAX H2R_TAGSTALL_MIN
H2R_TCPLAT_MAX H2R_TCPLAT_MIN H2R_TCPWR_MAX H2R_TCPWR_MIN
H2R_TCPPEND_MAX H2R_TCPPEND_MIN
H2R_WRREQ_XCC0 H2R_WRREQ_XCC1 H2R_WRREQ_XCC2 H2R_WRREQ_XCC3 H2R_WRREQ_XCC4 H2R_WRREQ_XCC5 H2R_WRREQ_XCC6 H2R_WRREQ_XCC7
H2R_WRLEVEL_XCC0 H2R_WRLEVEL_XCC1 H2R_WRLEVEL_XCC2 H2R_WRLEVEL_XCC3 H2R_WRLEVEL_XCC4 H2R_WRLEVEL_XCC5 H2R_WRLEVEL_XCC6 H2R_WRLEVEL_XCC7
H2R_WRREQ_CH0 H2R_WRREQ_CH1 H2R_WRREQ_CH2 H2R_WRREQ_CH3 H2R_WRREQ_CH4 H2R_WRREQ_CH5 H2R_WRREQ_CH6 H2R_WRREQ_CH7 H2R_WRREQ_CH8 H2R_WRREQ_CH9 H2R_WRREQ_CH10 H2R_WRREQ_CH11 H2R_WRREQ_CH12 H2R_WRREQ_CH13 H2R_WRREQ_CH14 H2R_WRREQ_CH15
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_WAVES
SETS
cat $OUT/passes.log | tail -12
du -sh $OUT
