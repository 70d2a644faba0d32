#!/bin/bash
# Placement classes under hardware counters: tools/placement_counters.py once plain (timings), then once per counter set under
# rocprofv3 --kernel-trace --pmc (counters in their own runs, never with the trace domains gpurun refuses).  Run on the GPU box:
#   gpurun --timeout 2400 -- 'bash tools/placement_counters.sh'
# Output under gpurun_out/r06_placement/; tools/placement_counters_parse.py turns it into profiles/r06_placement_counters.txt.
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
OUT=$ROOT/gpurun_out/r06_placement
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== kfd topology mem_banks"; for f in /sys/class/kfd/kfd/topology/nodes/*/mem_banks/*/properties; do echo "-- $f"; cat $f; done
  echo "== kfd node properties (gpu nodes)"; for f in /sys/class/kfd/kfd/topology/nodes/*/properties; do echo "-- $f"; grep -E "simd_count|mem_banks_count|caches_count|array_count|num_xcc|max_waves|gfx_target|local_mem_size|num_sdma" $f; done
  echo "== debugfs"; ls /sys/kernel/debug/dri/ 2>&1 | head; ls /sys/kernel/debug/dri/*/ 2>&1 | head -60
  echo "== amdgpu module params"; for p in vm_fragment_size vm_block_size vm_size mtype_local; do echo "$p = $(cat /sys/module/amdgpu/parameters/$p 2>&1)"; done
  echo "== memory partition"; cat /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition 2>&1
  rocm-smi --showmemuse --showmeminfo vram 2>&1 | head -20
} > $OUT/sysinfo.txt 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
( time python tools/placement_counters.py ) > $OUT/plain.log 2>&1
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  d=$OUT/pass$i
  rm -rf $d; mkdir -p $d
  echo "$SET" > $d/counters.txt
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $SET -d $d -o run --output-format csv -- python $ROOT/tools/placement_counters.py ) > $d/run.log 2>&1
  echo "pass $i ($SET): rc $?" >> $OUT/passes.log
  # keep the merged output small: only the dispatches of the two kernels
  for f in $(find $d -name "*counter_collection.csv"); do
    head -1 $f > $f.small; grep -E "lookup_fill_kernel|trace_kernel" $f >> $f.small; mv $f.small $f
  done
  for f in $(find $d -name "*kernel_trace.csv"); do
    head -1 $f > $f.small; grep -E "lookup_fill_kernel|trace_kernel" $f >> $f.small; mv $f.small $f
  done
  find $d -name "*agent_info.csv" -delete
done <<'SETS'
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum
TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum
TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum TCC_EA0_WRREQ_WRITE_IO_32B_sum TCC_EA0_WR_UNCACHED_32B_sum
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL
TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_WRITE_sum TCC_NORMAL_WRITEBACK_sum
TCC_EA0_ATOMIC_sum TCC_PROBE_sum TCC_BUBBLE_sum TCC_STREAMING_REQ_sum
SETS
ls -R $OUT | head -80
du -sh $OUT
