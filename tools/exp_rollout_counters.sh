#!/bin/bash
# SQ / memory counters of k_rollout:  tools/exp_rollout_counters.sh <tag> <iid|sorted>
tag="$1"; order="${2:-iid}"
for grp in "a:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "b:SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_BRANCH" "c:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "d:FETCH_SIZE" "e:SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" ; do
  name=${grp%%:*}; ctrs=${grp#*:}
  timeout -k 10 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $PWD/gpurun_out/roc_${tag}_$name -- python3 tools/bench_rollout.py 1048576 $order > gpurun_out/roc_${tag}_$name.log 2>&1; echo "pmc $tag $name exit $?"
  python tools/rocprof_summary.py gpurun_out/roc_${tag}_$name > gpurun_out/roc_${tag}_$name.summary.txt 2>&1
  find gpurun_out/roc_${tag}_$name -name "*.csv" -size +2M -delete
  grep k_rollout gpurun_out/roc_${tag}_$name.summary.txt | cut -c1-900
done
