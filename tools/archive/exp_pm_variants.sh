#!/bin/bash
# posterior-mean reward step time per build variant:  tools/exp_pm_variants.sh "flagsA;flagsB;..."
IFS=';' read -ra FL <<< "$1"
for flags in "${FL[@]}"; do
  echo "=== W2A_CXXFLAGS=$flags" | tee -a gpurun_out/pm_variants.log
  W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
  timeout -k 10 300 python tools/exp_posterior.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/pm_variants.log
done
python -c "from weather2alert_amd import build; build.build_lib(force=True)"
