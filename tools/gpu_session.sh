#!/bin/bash
# One GPU-box session of round 2: tests, then kernel A/B. Outputs under gpurun_out/. usage: tools/gpu_session.sh <steps...>
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m weather2alert_amd.build > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
# toolchain of this box, kept with the evidence (tools/collect_profiles.py copies it)
{ echo "# hipcc --version"; hipcc --version 2>&1; echo "# /opt/rocm/.info/version"; cat /opt/rocm/.info/version 2>/dev/null;
  echo "# rocminfo (gfx, CUs)"; rocminfo 2>/dev/null | grep -E "Marketing Name|Name: +gfx|Compute Unit" | sort | uniq -c | head -8;
  echo "# torch"; python -c "import torch; print(torch.__version__, torch.version.hip)"; } > gpurun_out/toolchain.txt 2>&1
for step in "$@"; do
  case $step in
    tests)
      timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1
      echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log ;;
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log ;;
    kernels)
      timeout -k 10 600 python tools/exp_step_kernels.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_step_kernels.log ;;
    ab)
      for flags in "-DW2A_S64_MIN_WAVES=5" "-DW2A_S64_MIN_WAVES=3" "-DBLOCK=128" "-DBLOCK=64"; do
        echo "=== W2A_CXXFLAGS=$flags" | tee -a gpurun_out/ab_step64.log
        W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
        timeout -k 10 300 python tools/exp_step_kernels.py --quick --kernels auto 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab_step64.log
      done
      python -c "from weather2alert_amd import build; build.build_lib(force=True)" ;;
    ab2)
      # AB_FLAGS="flagsA;flagsB;..."  -> bench.py kernel times (iid / always-alert / sorted) per build
      IFS=';' read -ra FL <<< "$AB_FLAGS"
      for flags in "${FL[@]}"; do
        echo "=== W2A_CXXFLAGS=$flags" | tee -a gpurun_out/ab2.log
        W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
        timeout -k 10 300 python bench.py --no-cpu-baseline --steps 612 ${AB_BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('iid %.2f us  always-alert %.2f us  sorted %.2f us  e2e %.2f G/s' % (d['roofline']['avg_launch_us'], d['always_alert_policy']['kernel_us'], d['sorted_episode_order']['kernel_us'], d['value']/1e9))" | tee -a gpurun_out/ab2.log
      done
      python -c "from weather2alert_amd import build; build.build_lib(force=True)" ;;
    benchall)
      for w in configs1 configs3; do
        timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1; echo "bench $w exit $?"; tail -1 gpurun_out/bench_$w.log | cut -c1-300
      done ;;
    gloo2)
      # two ranks sharing the one GPU over gloo: the whole multi-rank path of bench.py except RCCL itself
      timeout -k 10 300 python bench.py --gpus 2 --backend gloo --num-envs 262144 --steps 306 --no-cpu-baseline > gpurun_out/bench_gloo2.log 2>&1; echo "gloo2 exit $?"; tail -1 gpurun_out/bench_gloo2.log | cut -c1-400 ;;
    nsweep)
      for n in 16384 65536 131072 262144 524288; do
        timeout -k 10 300 python tools/exp_step_kernels.py --quick --num-envs $n --tag "n=$n " 2>&1 | grep -v amdgpu.ids | grep "random" | tee -a gpurun_out/nsweep.log
      done ;;
    rollout)
      timeout -k 10 300 python tools/bench_rollout.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_rollout.log ;;
    bench)
      timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log | cut -c1-3000 ;;
    benchab)
      for k in auto classic; do
        timeout -k 10 300 python bench.py --step-kernel $k --no-extras --no-cpu-baseline > gpurun_out/bench_$k.log 2>&1
        echo "bench $k exit $?"; tail -1 gpurun_out/bench_$k.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
      done ;;
    prof:*)
      # prof:<workload>[:sorted]  -- kernel trace of bench.py + PMC passes of tools/pmc_probe.py (one pass per counter
      # group, never combined with trace domains other than --kernel-trace)
      IFS=: read -r _ w order <<< "$step"
      tag=$w; extra=""; bextra=""
      if [ "$order" == "sorted" ]; then tag=${w}_sorted; extra="--episode-order sorted"; bextra="--episode-order sorted"; fi
      if [ "${order%%_*}" == "pm" ]; then
        # prof:<workload>:pm_<vector|matrix|matrix_i8>: issue counters (MFMA instructions and busy cycles, VALU, waits, LDS)
        # and fabric bytes of one posterior-mean reward kernel of the library, selected at run time
        pmk=${order#pm_}; [ "$pmk" == "pm" ] && pmk=vector
        tag=${w}_pm_$pmk
        rm -rf gpurun_out/prof_*_$tag gpurun_out/prof_*_$tag.summary.json gpurun_out/prof_*_$tag.summary.txt gpurun_out/prof_*_$tag.probe.json
        for grp in "mfma:SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS" "sq2:SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
          name=${grp%%:*}; ctrs=${grp#*:}
          timeout -k 10 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $PWD/gpurun_out/prof_${name}_$tag -- python3 tools/pmc_probe.py --workload $w --reward-mode posterior_mean --pm-kernel $pmk --steps 12 > gpurun_out/prof_${name}_$tag.log 2>&1; echo "prof_$name $tag exit $?"
          cp gpurun_out/pmc_probe_$tag.json gpurun_out/prof_${name}_$tag.probe.json
          python tools/rocprof_summary.py gpurun_out/prof_${name}_$tag --json gpurun_out/prof_${name}_$tag.summary.json > gpurun_out/prof_${name}_$tag.summary.txt 2>&1
          find gpurun_out/prof_${name}_$tag -name "*.csv" -size +2M -delete
          grep k_posterior gpurun_out/prof_${name}_$tag.summary.txt | cut -c1-700
        done
        continue
      fi
      R=$PWD
      rm -rf gpurun_out/prof_*_$tag gpurun_out/prof_*_$tag.summary.json gpurun_out/prof_*_$tag.summary.txt gpurun_out/prof_*_$tag.probe.json
      timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt_$tag -- python3 bench.py --workload $w $bextra --no-cpu-baseline --no-extras > gpurun_out/prof_kt_$tag.log 2>&1; echo "prof_kt $tag exit $?"
      for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "sq2:SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
        name=${grp%%:*}; ctrs=${grp#*:}
        timeout -k 10 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/prof_${name}_$tag -- python3 tools/pmc_probe.py --workload $w $extra > gpurun_out/prof_${name}_$tag.log 2>&1; echo "prof_$name $tag exit $?"
        cp gpurun_out/pmc_probe_$tag.json gpurun_out/prof_${name}_$tag.probe.json  # which kernel sources this pass ran on
      done
      for d in prof_kt prof_fetch prof_write prof_tcc prof_sq prof_sq2; do
        python tools/rocprof_summary.py gpurun_out/${d}_$tag --json gpurun_out/${d}_$tag.summary.json > gpurun_out/${d}_$tag.summary.txt 2>&1
        find gpurun_out/${d}_$tag -name "*.csv" -size +2M -delete
      done
      head -3 gpurun_out/prof_kt_$tag.summary.txt | cut -c1-300 ;;
  esac
done
