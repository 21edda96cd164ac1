#!/bin/bash
# A/B of build variants of the int8 matrix-core posterior-mean kernel: tools/exp_pm_i8_variants.sh "flagsA;flagsB;..."
IFS=';' read -ra FL <<< "$1"
for flags in "${FL[@]}"; do
  echo "=== W2A_CXXFLAGS=$flags" | tee -a gpurun_out/pm_i8_variants.log
  W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
  timeout -k 10 300 python tools/exp_pm_kernels.py --kernels ${PM_KERNELS:-matrix_i8} ${PM_ARGS:-} 2>&1 | grep -v amdgpu | tee -a gpurun_out/pm_i8_variants.log
done
python -c "from weather2alert_amd import build; build.build_lib(force=True)"
