#!/bin/bash
# One GPU-box session of round 2: tests, then kernel A/B. Outputs under gpurun_out/. usage: tools/gpu_session.sh <steps...>
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m weather2alert_amd.build > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
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
    bench)
      timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log | cut -c1-1500 ;;
  esac
done
