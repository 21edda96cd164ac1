#!/bin/bash
# SQ counters of a posterior-mean reward kernel (all three are in the one library):
#   tools/exp_pm_counters.sh "<W2A_CXXFLAGS>" <tag> [vector|matrix|matrix_i8] [workload]
flags="$1"; tag="$2"; pmk="${3:-vector}"; wl="${4:-configs3}"
W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
for grp in "a:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "b:SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_F64" "c:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"; do
  name=${grp%%:*}; ctrs=${grp#*:}
  timeout -k 10 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $PWD/gpurun_out/pmc_${tag}_$name -- python3 tools/pmc_probe.py --workload $wl --reward-mode posterior_mean --pm-kernel $pmk --steps 6 > gpurun_out/pmc_${tag}_$name.log 2>&1; echo "pmc $tag $name exit $?"
  python tools/rocprof_summary.py gpurun_out/pmc_${tag}_$name > gpurun_out/pmc_${tag}_$name.summary.txt 2>&1
  find gpurun_out/pmc_${tag}_$name -name "*.csv" -size +2M -delete
  grep k_posterior gpurun_out/pmc_${tag}_$name.summary.txt | cut -c1-900
done
python -c "from weather2alert_amd import build; build.build_lib(force=True)"
