#!/bin/bash
# on-device rollout time per build variant:  tools/exp_rollout_variants.sh "flagsA;flagsB;..."
IFS=';' read -ra FL <<< "$1"
for flags in "${FL[@]}"; do
  echo "=== W2A_CXXFLAGS=$flags" | tee -a gpurun_out/rollout_variants.log
  W2A_CXXFLAGS="$flags" python -c "from weather2alert_amd import build; build.build_lib(force=True)" || exit 1
  timeout -k 10 300 python tools/bench_rollout.py 2>&1 | grep "rollout" | tee -a gpurun_out/rollout_variants.log
done
python -c "from weather2alert_amd import build; build.build_lib(force=True)"
